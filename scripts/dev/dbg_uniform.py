import sys
sys.path[:0]=['flow-rl_amd','oracle','tests']
import numpy as np, torch, threefry_oracle as tf
from flowrl_hip import random as fr
key=tf.prng_key(20240); kt=torch.from_numpy(key).cuda()
got=fr.uniform(kt,(1000,1),1e-3,0.9946).cpu().numpy(); ref=tf.uniform(key,(1000,1),1e-3,0.9946)
d=got.view(np.int32)-ref.view(np.int32)
print("mismatch", (d!=0).sum(), "max ulp", np.abs(d).max(), got[:3].ravel(), ref[:3].ravel())
b=tf.random_bits(key,(1000,1))
f=((b>>np.uint32(9))|np.uint32(0x3F800000)).view(np.float32)-np.float32(1)
lo=np.float32(1e-3); hi=np.float32(0.9946)
fma=np.maximum(lo,(f.astype(np.float64)*np.float64(hi-lo)+np.float64(lo)).astype(np.float32))
print("matches fma emulation:", np.array_equal(got,fma))
gb=fr.bits(kt,(1000,1)).cpu().numpy(); print("bits equal", np.array_equal(gb,b))
