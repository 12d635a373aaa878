"""Whole-image inference throughput (SURVEY.md 8f N1 x N3): a PaviaU-sized scene (610 x 340 pixels) classified straight
from its cube by cmlpl_infer_cube, against the materialised-patch path (cmlpl_extract_patches into a recycled buffer +
the eval forward of cmlpl_basenet2_fwd), with the algorithmic FLOPs of the forward for the roofline fraction.
    python scripts/bench_infer.py [B2|B4|B5] [rows cols]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cmlpl_amd.infer import infer_cube
from cmlpl_amd.models import BaseNet2
from cmlpl_amd.patches import extract_patches

SHAPES = {"B2": (103, 11, 11, 103, 9), "B4": (200, 11, 11, 200, 16), "B5": (48, 15, 15, 48, 20)}
name = sys.argv[1] if len(sys.argv) > 1 else "B2"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 610
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 340
C, H, W, bands, K = SHAPES[name]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1088)
cube = torch.randn(rows, cols, C, device=dev, generator=g)
X = torch.randn(rows * cols, bands, device=dev, generator=g)
net = BaseNet2(num_features=bands, dropout=0.8, num_classes=K, in_channels=C, window=H).to(dev).eval()
n = rows * cols
H2, W2 = H // 2, W // 2
flop = 2.0 * (H * W * 64 * C + (2 * H2) * (2 * W2) * 64 * 576 + (2 * (H2 // 2)) * (2 * (W2 // 2)) * 64 * 576) + 2.0 * 1024 * bands
print(f"{name}: scene {rows} x {cols} x {C} ({n} pixels, cube {cube.numel() * 4 / 1e6:.1f} MB; the materialised patches would be "
      f"{n * C * H * W * 4 / 1e9:.2f} GB), window {H} x {W}, {flop / 1e6:.2f} MFLOP per pixel (convolutions + spectral GEMM)")


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]


for chunk in (16384, 65536, n):
    best, med = timed(lambda: infer_cube(net, cube, X, chunk=chunk))
    print(f"  cube path, {chunk:7d} pixels per launch: {med * 1e3:8.2f} ms (best {best * 1e3:.2f}) = {n / med / 1e6:6.2f} M pixels/s = "
          f"{n * flop / med / 1e12:6.1f} TFLOP/s = {n * flop / med / 1e12 / (2500 / 6):.3f} of the split-bf16 MFMA ceiling")

buf = torch.empty(8192, C, H, W, device=dev)
idx = torch.arange(n, device=dev)


def patch_path():
    out = []
    with torch.no_grad():
        for o in range(0, n, 8192):
            m = min(8192, n - o)
            XP = extract_patches(cube, idx[o:o + m], H, out=buf[:m])
            z, _ = net(XP, X[o:o + m])
            out.append(z.argmax(1))
    return torch.cat(out)


best, med = timed(patch_path, reps=3)
print(f"  patch path (extract 8192 patches into a recycled buffer + eval forward + argmax): {med * 1e3:8.2f} ms = {n / med / 1e6:6.2f} M pixels/s")
a = infer_cube(net, cube, X)
b = patch_path()
print(f"  labels equal on {float((a == b).float().mean()) * 100:.3f} % of the pixels (ties at rounding level may differ)")
