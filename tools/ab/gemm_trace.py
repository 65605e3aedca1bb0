"""Per-block phase anatomy of gemm_v3 (needs tools/ab/libTRACE.so = build_variant.sh TRACE -DSIDLSG_EXP_TRACE).
phases (100 MHz wall clock): 0 entry -> 1 first K-tile landed -> 2 K loop done -> 3 epilogue math + LDS image -> 4 stores issued
-> 5 stores acknowledged."""
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
for M, N, K in ((65536, 2560, 320), (65536, 320, 320), (16384, 5120, 640), (65536, 320, 1280)):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    nblk = ((M + 127) // 128) * ((N + 159) // 160) * 8
    tr = torch.zeros(nblk * 8, device=dev, dtype=torch.int64)
    for _ in range(3):
        ops.gemm(a, w, out=c)
    torch.cuda.synchronize()
    assert dll.sidlsg_exp_set_trace(ctypes.c_void_p(tr.data_ptr())) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(a, w, out=c); e1.record()
    torch.cuda.synchronize()
    dll.sidlsg_exp_set_trace(ctypes.c_void_p(0))
    t = tr.cpu().numpy().reshape(-1, 8)
    t = t[t[:, 0] != 0]
    ts = t[:, :6].astype(np.float64) * 0.01          # us
    t0 = ts[:, 0].min()
    d = np.diff(ts, axis=1)
    print(f'{M}x{N}x{K}: {len(t)} blocks, kernel {e0.elapsed_time(e1) * 1e3:.1f} us (event), span {ts[:, 5].max() - t0:.1f} us')
    print('   phase means us: load0 %.2f | kloop %.2f | epi %.2f | store-issue %.2f | store-ack %.2f | total %.2f' % (*d.mean(0), (ts[:, 5] - ts[:, 0]).mean()))
    print('   phase p90   us: load0 %.2f | kloop %.2f | epi %.2f | store-issue %.2f | store-ack %.2f' % tuple(np.percentile(d, 90, axis=0)))
    hw = t[:, 7]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)     # xcc, se, cu
    ncu = len(np.unique(cu))
    start = ts[:, 0] - t0
    print(f'   distinct (xcc,se,cu): {ncu}; blocks/CU {len(t) / ncu:.1f}; start-time quartiles us {np.percentile(start, [25, 50, 75, 100]).round(1)}')
    # concurrency: average number of resident blocks
    print(f'   mean resident blocks {(ts[:, 5] - ts[:, 0]).sum() / (ts[:, 5].max() - t0):.0f}')
