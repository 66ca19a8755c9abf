import torch, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tspo_amd import preprocess as P
dev = "cuda"
for (H, W, N) in ((720, 1280, 256), (360, 640, 512), (224, 224, 512)):
    x = torch.randint(0, 256, (N, H, W, 3), dtype=torch.uint8, device=dev)
    for _ in range(2):
        y = P.preprocess_frames(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        y = P.preprocess_frames(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{H}x{W} x{N}: {dt*1e3:.2f} ms -> {N/dt:.0f} frames/s, in {x.numel()/dt/1e9:.0f} GB/s, out {tuple(y.shape)} {y.dtype}")
