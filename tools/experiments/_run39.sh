# fused GEGLU backward (FF-out data gradient + GEGLU derivative): tests, kernel micro-benchmark, step A/B
set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "feed_forward or geglu" > gpurun_out/r39_tests.log 2>&1
tail -n 15 gpurun_out/r39_tests.log
python - <<'P' 2>&1 | tee gpurun_out/r39_micro.log
import torch, sys
sys.path.insert(0, '.')
from sid_lsg_amd import ops
from sid_lsg_amd._lib import lib
from tools.bench_kernels import timeit, r
for M, C in ((65536, 320), (32768, 320), (16384, 640), (4096, 1280)):
    F = 4 * C
    dout, w2t, h = r(M, C), r(F, C, scale=0.02), r(M, 2 * F)
    dh = torch.empty_like(h); dy = torch.empty(M, F, device='cuda', dtype=torch.bfloat16)
    t_f = timeit(lambda: lib.sidlsg_gemm_geglu_bwd_bf16(dout.data_ptr(), C, w2t.data_ptr(), h.data_ptr(), dh.data_ptr(), 2 * F, M, F, C, ops._s()), iters=20)
    t_g = timeit(lambda: ops.gemm(dout, w2t, out=dy), iters=20)
    t_e = timeit(lambda: lib.sidlsg_geglu_bwd(h.data_ptr(), dy.data_ptr(), dh.data_ptr(), M, F, ops._s()), iters=20)
    print(f'M{M} C{C}: fused {t_f*1e6:7.1f} us | gemm {t_g*1e6:7.1f} + geglu_bwd {t_e*1e6:7.1f} = {(t_g+t_e)*1e6:7.1f} us')
P
timeout 300 python tools/bench_kernels.py conv gemm 2>/dev/null | grep -E "aggregate" | tee -a gpurun_out/r39_micro.log
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-kernel-timing --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['loss_check'])" >> gpurun_out/r39_ab.log; }
rm -f gpurun_out/r39_ab.log
run SIDLSG_GEMM_GEGLU_BWD=1; run SIDLSG_GEMM_GEGLU_BWD=0; run SIDLSG_GEMM_GEGLU_BWD=1; run SIDLSG_GEMM_GEGLU_BWD=0; run SIDLSG_GEMM_GEGLU_BWD=1; run SIDLSG_GEMM_GEGLU_BWD=0
cat gpurun_out/r39_ab.log
