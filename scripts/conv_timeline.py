"""Ablation aid: per-workgroup phase timeline of conv3x3_kernel (needs the CMLPL_ABL=9 build of the library).
   CMLPL_LIB=cmlpl_amd/libabl9.so python scripts/conv_timeline.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cmlpl_amd import TrainEngine, NetShape, HyperParams, _lib

C = int(sys.argv[1]) if len(sys.argv) > 1 else 103          # bands: 103 = B2, 200 = B4
K = int(sys.argv[2]) if len(sys.argv) > 2 else 9
BT = int(sys.argv[3]) if len(sys.argv) > 3 else 128          # rows per network: 64 64 = a rank's shard of configs[2] at 8 GPUs
BTU = int(sys.argv[4]) if len(sys.argv) > 4 else 128
WIN = int(os.environ.get("CMLPL_TL_WIN", "11"))           # window side: 11 = B2 / B4, 15 = B5
shape = NetShape(C, WIN, WIN, C, K)
eng = TrainEngine(shape, BT, BTU, HyperParams(), device="cuda:0", seed=1)
eng.init_params_default(1)          # (zero parameters give zero images: the two-piece loop would not be taken)
g = torch.Generator(device="cuda:0").manual_seed(0)
XPl = torch.randn(BT, C, WIN, WIN, device="cuda:0", generator=g)
XPu = torch.randn(BTU, C, WIN, WIN, device="cuda:0", generator=g)
Xl = torch.randn(BT, C, device="cuda:0", generator=g)
Xu = torch.randn(BTU, C, device="cuda:0", generator=g)
Y = torch.randint(0, K, (BT,), device="cuda:0", generator=g)
lib = _lib.load()
# (general path, e.g. 20x20 windows: build the library with ABL_FLAGS=-DCMLPL_STAMP_MIN_H=20 to see conv1's launches
#  instead of conv2's on the pooled map, which come later and overwrite the slots)
for i in range(20):
    eng.step(XPl, Xl, Y, XPu, Xu, 1, i)
torch.cuda.synchronize()
from cmlpl_amd.build_ext import embedded_hash, source_hash
_h = embedded_hash(os.environ.get("CMLPL_LIB"))
print(f"timeline library built from sources {_h}; sources here {source_hash()}" + ("" if _h == source_hash() else "   ** STALE BUILD: run scripts/build_abl.sh 9 **"))
buf = np.zeros((3, 2048, 16), dtype=np.uint64)
rc = lib.cmlpl_abl_read_stamps(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
wbuf = np.zeros((2048, 16), dtype=np.uint64)          # the weight-gradient kernels live in their own translation unit
rc = lib.cmlpl_abl_read_wstamps(wbuf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0
buf[2] = wbuf
NWG = 2 * (BT + BTU)
print(f"{BT}+{BTU} rows, {NWG} sample-net workgroups, CMLPL_KS8={os.environ.get('CMLPL_KS8', '(planner)')}")
for mode, name, nwg in ((0, "fwd", NWG), (1, "dgrad", NWG), (2, "wgrad(conv1 workgroups of the pair launch)", 192)):
    t = buf[mode, :nwg, :4].astype(np.int64)
    full = buf[mode, :nwg, :].astype(np.int64)
    full = full[t[:, 0] > 0]
    t = t[t[:, 0] > 0]
    if mode == 0 and os.environ.get("CMLPL_LIB", "").endswith("libabl25.so"):
        f = (full - full[:, :1]) / 100.0
        print(f"fwd tail detail (CMLPL_ABL=25; us from workgroup start): pooled {f[:,7].mean():.2f}  up-front loads issued + dropout "
              f"formed {f[:,4].mean():.2f}  barrier passed {f[:,5].mean():.2f}  planes split + barrier {f[:,6].mean():.2f}  conv2 MFMAs done "
              f"{f[:,8].mean():.2f}  conv2 epilogue done {f[:,12].mean():.2f}  head barrier passed {f[:,13].mean():.2f}  dropout row "
              f"done {f[:,9].mean():.2f}  end {f[:,3].mean():.2f}")
    elif mode == 0 and full[:, 4].max() > 0:
        f = (full - full[:, :1]) / 100.0
        c3 = f"{f[:,13].mean():.2f}" if full[:, 13].min() > 0 else "n/a (fewer than four 16-band chunks)"
        print(f"fwd fused prologue (us from workgroup start): loads issued + noise formed {f[:,4].mean():.2f}  chunk 0 in LDS "
              f"{f[:,12].mean():.2f}  chunk 3 in LDS {c3}  conv0 MFMAs done {f[:,5].mean():.2f}  "
              f"image zeroed {f[:,6].mean():.2f}  a0 written / stage end {f[:,1].mean():.2f}")
    if mode == 0 and full[:, 9].max() > 0:
        f = (full - full[:, :1]) / 100.0
        print(f"fwd tail (us from workgroup start): taps end {f[:,2].mean():.2f}  pooled {f[:,7].mean():.2f}  conv2 MFMAs done "
              f"{f[:,8].mean():.2f}  dropout row done {f[:,9].mean():.2f}  end {f[:,3].mean():.2f}")
    if mode == 0 and full[:, 11].max() > 0:
        f = (full - full[:, :1]) / 100.0
        print(f"fwd detail: tap loop entered {f[:,14].mean():.2f}  wave-0 taps done {f[:,15].mean():.2f}  all waves done {f[:,2].mean():.2f}  "
              f"relu written {f[:,10].mean():.2f}  barrier passed {f[:,11].mean():.2f}  pooled {f[:,7].mean():.2f}")
    if mode == 1 and full[:, 11].max() > 0:
        f = (full - full[:, :1]) / 100.0
        print(f"bwd head (us from workgroup start): head done {f[:,10].mean():.2f}  conv2 dgrad done {f[:,11].mean():.2f}  stage end "
              f"{f[:,1].mean():.2f}  taps end {f[:,2].mean():.2f}  slab issued+da0 {f[:,12].mean():.2f}  noise+landed {f[:,13].mean():.2f}  "
              f"end {f[:,3].mean():.2f}")
    if mode == 1 and full[:, 5].max() > 0:
        f = (full - full[:, :1]) / 100.0
        print(f"bwd conv2 data gradient (us from workgroup start): head done {f[:,10].mean():.2f}  barrier passed {f[:,4].mean():.2f}  dz2 staged + "
              f"barrier {f[:,5].mean():.2f}  MFMA loop done {f[:,6].mean():.2f}  residual read + barrier {f[:,7].mean():.2f}  exchange + barrier "
              f"{f[:,8].mean():.2f}  dp1 written + barrier {f[:,11].mean():.2f}")
    t0 = t[:, 0].min()
    us = (t - t0) / 100.0
    print(f"{name}: start skew  mean {us[:,0].mean():.2f}  max {us[:,0].max():.2f} us")
    print(f"{name}: stage  {np.mean(us[:,1]-us[:,0]):.2f} us   taps {np.mean(us[:,2]-us[:,1]):.2f} us   "
          f"epilogue {np.mean(us[:,3]-us[:,2]):.2f} us   last end {us[:,3].max():.2f} us")
    print(f"   per-WG percentiles of (taps): {np.percentile(us[:,2]-us[:,1],[5,50,95])}")
    print(f"   per-WG percentiles of (stage): {np.percentile(us[:,1]-us[:,0],[5,50,95])}")
    print(f"   per-WG percentiles of (end): {np.percentile(us[:,3],[5,50,95])}")
