#!/bin/bash
# round 6, call 38: the fused finish of narrow chains and launch 3 capped at three wavefronts per SIMD: parity, A/B (fused finish on / off; launch 3 at 1 / 2 / 3 wavefronts per SIMD, libraries built here)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r06_call38
mkdir -p $OUT
cd $REPO
timeout 2000 python -m pytest tests/test_gpu_narrow_chain.py tests/test_gpu_transcript_coop.py tests/test_gpu_rangeproof.py tests/test_gpu_transcript_stop.py tests/test_gpu_combine.py tests/test_gpu_pool.py tests/test_gpu_coalesce_shapes.py tests/test_gpu_concurrency.py tests/test_gpu_stress_mixed.py -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -5 $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
LIB=$REPO/bulletproofs_amd/csrc
g++ -O2 -std=c++17 -pthread -I $REPO/include $REPO/tools/combine_rate.cpp -L $LIB -lbpgpu -o /tmp/combine_rate || exit 1
INP=$REPO/bench_data/combine_rate_inputs.bin
export BP_LANES=8 BP_W=16 GPU_MAX_HW_QUEUES=16
run() { # file libdir env-string mode...
  local f=$1 l=$2 e=$3; shift 3
  env LD_LIBRARY_PATH=$l $e /tmp/combine_rate $INP 1.0 "$@" 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$e', '$*', 'rate', d['rate_per_s'], 'lat', d['lat_ms'], 'per chain', d['proofs_per_chain'], 'mismatches', d['mismatches'], 'errors', d['errors'])" >> $OUT/$f
}
for rep in 1 2 3; do
  for e in "BPGPU_NARROW_FUSED_FINISH=0" "BPGPU_NARROW_FUSED_FINISH=1"; do
    run fused_finish_ab.txt $LIB "$e" threads 1
    run fused_finish_ab.txt $LIB "$e" threads 16
    run fused_finish_ab.txt $LIB "$e" threads 64
    run fused_finish_ab.txt $LIB "$e" threads 256
    run fused_finish_ab.txt $LIB "$e" tickets 16 128
  done
done
cat $OUT/fused_finish_ab.txt
for w in 1 2; do
  mkdir -p /tmp/var_w$w
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -DBP_STAGE3_WAVES=$w -c -o /tmp/var_w$w/k_rp34.o $LIB/k_rp34.hip || continue
  objs=$(ls $LIB/build/*.o | grep -v k_rp34.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/var_w$w/libbpgpu.so $objs /tmp/var_w$w/k_rp34.o || continue
done
for rep in 1 2 3; do
  for w in 1 2 3; do
    l=/tmp/var_w$w; [ $w = 3 ] && l=$LIB
    run stage3_waves_ab.txt $l "BP_STAGE3_WAVES_BUILD=$w" threads 64
    run stage3_waves_ab.txt $l "BP_STAGE3_WAVES_BUILD=$w" threads 256
    run stage3_waves_ab.txt $l "BP_STAGE3_WAVES_BUILD=$w" tickets 16 128
    run stage3_waves_ab.txt $l "BP_STAGE3_WAVES_BUILD=$w" tickets 16 512
  done
done
cat $OUT/stage3_waves_ab.txt
