import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ctypes as C, numpy as np, torch
from tspo_amd import _lib, preprocess as P, synth
H, W, T = 360, 640, 2
frames = torch.from_numpy(synth.uniform_u8((T, H, W, 3), 55 + H + W)).cuda()
hk, hb, vk, vb, ylo, nrows = P._tables(H, W, 224)
mf = P.mfma_h_tables(hk, hb)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
thk, thb, tvk, tvb = map(dev, (hk, hb, vk, vb))
mt, mb, mx = dev(mf[0]), dev(mf[1]), dev(mf[2])
def run(use):
    ws = torch.zeros(T * 3 * nrows * 224 + 256, dtype=torch.uint8, device="cuda")
    out = torch.empty((T, 3, 224, 224), dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = _lib.lib().tspo_preprocess_frames_ex(p(frames), 0, T, H, W, p(thk), p(thb), 224, hk.shape[1], p(tvk), p(tvb), 224, vk.shape[1],
        ylo, nrows, p(out), p(ws), ws.numel(), None, p(mt) if use else None, p(mb), p(mx), mf[3], mf[4])
    torch.cuda.synchronize(); assert rc == 0, _lib.lib().tspo_last_error()
    return ws[:T * 3 * nrows * 224].view(T, 3, nrows, 224).cpu().numpy()
import sys
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
if mode == "c128": frames.fill_(128)
if mode == "c200": frames.fill_(200)
if mode == "ramp": frames.copy_((torch.arange(W, device="cuda") % 251).to(torch.uint8)[None, None, :, None].expand(T, H, W, 3))
a, b = run(False), run(True)
bad = a != b
print("nkb", mf[3], "span", mf[4], "mismatch frac", bad.mean())
print("by channel", bad.mean((0, 2, 3)), "by x%16", bad.mean((0, 1, 2)).reshape(14, 16).mean(0).round(2))
print("by block", bad.mean((0, 1, 2)).reshape(14, 16).mean(1).round(2))
print("by row%32", bad.mean((0, 1, 3))[:64].round(2))
print(a[0, 0, 0, :24]); print(b[0, 0, 0, :24])
print(a[0, 1, 0, :24]); print(b[0, 1, 0, :24])
