mkdir -p gpurun_out/diag
T="tests/test_gpu_unet.py::test_glue_matches_reference_golden tests/test_gpu_unet.py::test_product_loop_matches_reference_golden"
for v in cur r5norm r5; do
  if [ $v = cur ]; then unset SIDLSG_LIB; else export SIDLSG_LIB=$PWD/tools/variants/libsidlsg_$v.so; fi
  python -m pytest $T -m gpu -q -s 2>&1 | grep -i "denoise tiny40 b1 kappa 1.0\|differs\|passed\|failed\|curve\|loss" | head -20 > gpurun_out/diag/$v.txt
done
unset SIDLSG_LIB
SIDLSG_GEMM_P8=0 python -m pytest $T -m gpu -q -s 2>&1 | grep -i "denoise tiny40 b1 kappa 1.0\|differs\|passed\|failed" | head -20 > gpurun_out/diag/cur_p8off.txt
python -m pytest tests/test_gpu_p8.py -m gpu -q 2>&1 | tail -3 > gpurun_out/diag/p8.txt
python tools/teacher_pass_graph.py > gpurun_out/diag/teacher_graph.txt 2>&1
python tools/gemm_shape_profile.py > gpurun_out/diag/gemm_shapes.txt 2>&1
python tools/gemm_shape_profile.py --entry sidlsg_conv3x3_bf16 > gpurun_out/diag/conv_shapes.txt 2>&1
head -5 gpurun_out/diag/*.txt
