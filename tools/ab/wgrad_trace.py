"""Per-block phase anatomy of wgrad_v2 (needs tools/ab/libTRACE.so).  0 entry -> 1 first stage landed -> 2 pixel loop done ->
3 result stores issued -> 4 stores acknowledged."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('SIDLSG_LIB', os.path.join(ROOT, 'tools', 'ab', 'libTRACE.so'))
sys.path.insert(0, ROOT)
import ctypes
import numpy as np
import torch
from sid_lsg_amd import ops
from sid_lsg_amd._lib import lib
dll = lib.load()
dev = torch.device('cuda:0')
BF = torch.bfloat16
def run(name, fn, nblk_max=1 << 16):
    tr = torch.zeros(nblk_max * 8, device=dev, dtype=torch.int64)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    assert dll.sidlsg_exp_set_trace(ctypes.c_void_p(tr.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    dll.sidlsg_exp_set_trace(ctypes.c_void_p(0))
    t = tr.cpu().numpy().reshape(-1, 8); t = t[t[:, 0] != 0]
    ts = t[:, :5].astype(np.float64) * 0.01
    d = np.diff(ts, axis=1)
B = 16
for M, N, K in ((B * 4096, 320, 320), (B * 4096, 2560, 320), (B * 1024, 640, 2560), (B * 256, 1280, 1280)):
    dy = torch.randn(M, N, device=dev).to(BF); a = torch.randn(M, K, device=dev).to(BF)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    run(f'dense {M}x{N}x{K}', lambda: lib.sidlsg_wgrad_bf16(dy.data_ptr(), N, a.data_ptr(), K, dw.data_ptr(), db.data_ptr(), M, N, K, ops._s()))
for H, cin, cout in ((64, 320, 320), (32, 640, 640), (16, 1280, 1280)):
    x = torch.randn(B, H, H, cin, device=dev).to(BF); dy = torch.randn(B, H, H, cout, device=dev).to(BF)
    dw = torch.zeros(cout, 9 * cin, device=dev); db = torch.zeros(cout, device=dev)
    run(f'conv {H}x{H} {cin}->{cout}', lambda: lib.sidlsg_conv3x3_wgrad_bf16(dy.data_ptr(), cout, x.data_ptr(), cin, dw.data_ptr(), db.data_ptr(), B, H, H, cin, cout, 1, 0, ops._s()))
