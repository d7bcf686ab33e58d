// COMPILE SMOKE ONLY.  This is NOT jaxlib's xla/ffi/api/ffi.h (absent from this image, not fetchable): it is a minimal
// in-tree declaration of the handful of xla::ffi names flow-rl_amd/csrc/xla_ffi_shim.cc uses, with the shapes those
// names have in the XLA FFI API of jaxlib 0.5.x as the build's authors know it (Error / ErrorCode / ErrorOr, DataType,
// AnyBuffer, Result, RemainingArgs / RemainingRets, Dictionary, ScratchAllocator, PlatformStream, Ffi::Bind() with
// Ctx / RemainingArgs / RemainingRets / Attrs, XLA_FFI_DEFINE_HANDLER_SYMBOL).  tests/test_abi.py runs
// `g++ -fsyntax-only` on the shim against it: that catches typos, missing includes and arity / type mistakes in the
// shim; it proves nothing about jaxlib's real header, and nothing here is ever linked or shipped.
#pragma once
#include <cstddef>
#include <cstdint>
#include <optional>
#include <string>
#include <string_view>
#include <tuple>
#include <utility>

struct XLA_FFI_CallFrame;
struct XLA_FFI_Error;

namespace xla::ffi {

enum class ErrorCode { kOk, kInvalidArgument, kUnimplemented, kResourceExhausted, kInternal };
enum class DataType { F32, U32, S32, BF16, U8 };

class Error {
 public:
  Error() = default;
  Error(ErrorCode code, std::string message) : code_(code), message_(std::move(message)) {}
  static Error Success() { return Error(); }
  bool success() const { return code_ == ErrorCode::kOk; }
  bool failure() const { return !success(); }
 private:
  ErrorCode code_ = ErrorCode::kOk;
  std::string message_;
};

template <typename T>
class ErrorOr {
 public:
  ErrorOr(T v) : v_(std::move(v)) {}
  ErrorOr(Error) {}
  bool has_value() const { return v_.has_value(); }
  T& operator*() { return *v_; }
  const T& operator*() const { return *v_; }
  T* operator->() { return &*v_; }
  const T* operator->() const { return &*v_; }
 private:
  std::optional<T> v_;
};

template <typename T>
class Span {
 public:
  size_t size() const { return n_; }
  const T& back() const { return p_[n_ - 1]; }
  const T& operator[](size_t i) const { return p_[i]; }
 private:
  const T* p_ = nullptr;
  size_t n_ = 0;
};

class AnyBuffer {
 public:
  DataType element_type() const { return DataType::F32; }
  Span<int64_t> dimensions() const { return {}; }
  size_t element_count() const { return 0; }
  void* untyped_data() const { return nullptr; }
};

template <typename T>
class Result {
 public:
  T* operator->() { return &v_; }
  T& operator*() { return v_; }
 private:
  T v_;
};

class RemainingArgs {
 public:
  size_t size() const { return 0; }
  template <typename T> ErrorOr<T> get(size_t) const { return Error(); }
};

class RemainingRets {
 public:
  size_t size() const { return 0; }
  template <typename T> ErrorOr<Result<T>> get(size_t) const { return Error(); }
};

class Dictionary {
 public:
  template <typename T> ErrorOr<T> get(std::string_view) const { return Error(); }
};

class ScratchAllocator {
 public:
  std::optional<void*> Allocate(size_t, size_t = 1) { return std::nullopt; }
};

template <typename T> struct PlatformStream {};

namespace internal {
template <typename... Ts> struct Binding {
  template <typename T> Binding<Ts..., T> Ctx() const { return {}; }
  Binding<Ts..., ::xla::ffi::RemainingArgs> RemainingArgs() const { return {}; }
  Binding<Ts..., ::xla::ffi::RemainingRets> RemainingRets() const { return {}; }
  Binding<Ts..., ::xla::ffi::Dictionary> Attrs() const { return {}; }
};
template <typename T> struct Decoded { using type = T; };
template <typename T> struct Decoded<PlatformStream<T>> { using type = T; };
// the handler must be callable with the decoded binding types, in order, and return Error
template <typename Fn, typename... Ts>
constexpr bool Callable(const Binding<Ts...>&, Fn fn) {
  using R = decltype(fn(std::declval<typename Decoded<Ts>::type>()...));
  static_assert(std::is_same_v<R, Error>, "an FFI handler returns xla::ffi::Error");
  return true;
}
}  // namespace internal

struct Ffi {
  static internal::Binding<> Bind() { return {}; }
};

}  // namespace xla::ffi

#define XLA_FFI_DEFINE_HANDLER_SYMBOL(name, impl, binding)                                   \
  static const bool name##_binding_checked = ::xla::ffi::internal::Callable((binding), impl); \
  extern "C" XLA_FFI_Error* name(XLA_FFI_CallFrame*) { return nullptr; }
