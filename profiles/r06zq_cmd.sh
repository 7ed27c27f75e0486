# round 6, GPU call zq: `mc X.tla` moving from the device interpreter to generated code by itself, COLD cache (hipcc runs beside the search):
# the Michael-Scott queue model with counted pointers, N = 3, K = 4 (124.6 M states) and K = 5, three ways each: interpreter only
# ($TLAMC_AUTOJIT=0), the default, -jit (generated code from the first state, the build in front)
cd /root/repo; D=$PWD/gpurun_out/r06zq; mkdir -p $D
MC=tla_rust_amd/_build/mc
for K in 4 5; do
  printf 'CONSTANTS N = 3 K = %d Counted = TRUE\nINVARIANTS HeadLive TailLive PointersAreNodes TailAtMostOneBehind CountsGrow\n' $K > $D/k$K.cfg
  A=140000000; T=30; [ $K = 5 ] && A=700000000 && T=32
  for mode in interp auto jit; do
    export TLAMC_JIT_CACHE=$D/cache_${K}_$mode; rm -rf $TLAMC_JIT_CACHE
    opt=""; env_aj=1; [ $mode = interp ] && env_aj=0; [ $mode = jit ] && opt="-jit"
    s=$(date +%s.%N)
    TLAMC_AUTOJIT=$env_aj timeout 600 $MC specs/pluscal/ms_queue_counted.tla -config $D/k$K.cfg -tablelog2 $T -arena $A -chunk 2097152 -noprogress $opt > $D/k${K}_$mode.out 2> $D/k${K}_$mode.err; rc=$?
    e=$(date +%s.%N)
    echo "{\"K\": $K, \"mode\": \"$mode\", \"rc\": $rc, \"wall_s\": $(python -c "print(round($e-$s,2))"), \"counts\": \"$(grep 'states generated' $D/k${K}_$mode.out | tail -1)\", \"stderr\": \"$(grep -v amdgpu.ids $D/k${K}_$mode.err | tr '\n"' '  ' | cut -c1-200)\"}" | tee -a $D/autojit.jsonl
    rm -rf $TLAMC_JIT_CACHE
  done
done
