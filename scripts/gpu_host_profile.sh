#!/bin/bash
# Host time of one ss_ctx_observe call by segment (-DSS_AB library prebuilt in gpurun_in/: ss_ab_host_profile), per config;
# same-box A/B of the input fence of the overlap mode (SS_HIP_ALWAYS_FENCE=1 = round 4's behaviour: record + wait on every call)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/host_profile; mkdir -p $OUT; rm -f $OUT/host_profile.txt
cp sound-spaces_amd/csrc/libss_hip.so /tmp/libss_hip_product.so
cp gpurun_in/libss_hip_ab.so sound-spaces_amd/csrc/libss_hip.so
export SS_AB_HOST_PROFILE=1
for rep in 1 2; do
for cfg in cfg3 cfg1 headline; do
  for fence in 1 0; do
    if [ $fence = 1 ]; then export SS_HIP_ALWAYS_FENCE=1; else unset SS_HIP_ALWAYS_FENCE; fi
    echo "== $cfg always_fence=$fence rep=$rep" >> $OUT/host_profile.txt
    timeout 300 python bench.py --config $cfg --no-cpu-baseline --no-plugin-path --no-secondary --regions 5 --sustain 0 \
        2>> $OUT/host_profile.txt | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','host_us_per_call')}))" >> $OUT/host_profile.txt
  done
done
done
unset SS_HIP_ALWAYS_FENCE
cp /tmp/libss_hip_product.so sound-spaces_amd/csrc/libss_hip.so
grep -v amdgpu $OUT/host_profile.txt
timeout 600 python -m pytest tests/test_context.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
