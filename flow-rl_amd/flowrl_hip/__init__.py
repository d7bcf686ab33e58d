"""flowrl_hip — MI355X-native fused action sampler behind FlowRL's sampler entry points.

Drop-in surface (same names, argument order and return structure as the reference):
  flowrl_hip.flow.ddpm.DDPM.sample                        <- flowrl/flow/ddpm.py:173-221
  flowrl_hip.flow.continuous_ddpm.ContinuousDDPM.sample   <- flowrl/flow/continuous_ddpm.py:117-222
  flowrl_hip.flow.cnf.ContinuousNormalizingFlow.sample    <- flowrl/flow/cnf.py:113-141
  flowrl_hip.critic.critic_select / sample_and_select     <- flowrl/agent/offline/dql.py:130-162 (critic best-of-N)
  flowrl_hip.autograd.sample / apply / value_and_grad     <- jax.grad(loss_fn)(params) of module/model.py:78-86
All arithmetic runs in libflowrl_hip.so (hand-written HIP for gfx950) through the C ABI in
include/flowrl_hip.h; there is no CPU / eager fallback.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401
from . import random  # noqa: F401
from . import autograd  # noqa: F401  (torch.autograd bridge: jax.grad's place around sample / apply)
from . import model  # noqa: F401  (Model.apply_gradient: TrainState, adam, clip_by_global_norm)
from .module import (DDPMBackbone, FlowBackbone, ContinuousDDPMBackbone, MLP, ResidualMLP, Simba, PositionalEmbedding,  # noqa: F401
                     LearnableFourierEmbedding, init_params, tree_from_numpy, tree_to)
from ._lib import FlowRLHipError, FlowRLHipUnsupported  # noqa: F401
from .critic import Ensemblize, FusedSelect, ScalarCritic, critic_select, init_critic_params, sample_and_select  # noqa: F401


def last_kernels() -> str:
    """Names (and launch geometry) of the kernels the calling thread's last library call enqueued (frl_last_kernels)."""
    from . import _lib
    return _lib.load().frl_last_kernels().decode()
