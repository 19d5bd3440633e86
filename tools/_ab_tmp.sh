for i in 1 2; do
for v in "" _es; do
PNA_AMD_LIB_PATH=$PWD/pna_amd/lib/libpna_amd$v.so python tools/bench_train.py 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('lib[$v]', d['fwd_bwd_ms'], d['fwd_bwd_given_output_gradient_ms']); print({k[:50]:v for k,v in list(d['top_kernels_us'].items())[:3]})"
done; done
PNA_AMD_LIB_PATH=$PWD/pna_amd/lib/libpna_amd_es.so python -m pytest tests/test_gpu_backward.py tests/test_gpu_train_trace.py -q 2>&1 | grep -E "passed|failed" | tail -2
