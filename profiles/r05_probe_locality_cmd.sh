# NOT RUN in round 4 (no GPU-minutes left when it was written): the first GPU call of the next round.  The random-probe rate of the
# seen-set against the working set it lands in (L2 / Infinity Cache / HBM / beyond the translation caches' reach), first-time inserts,
# and the end-to-end rate of probing candidates that were first counting-sorted by table region — for the 8 GiB table of atomic_add
# N = 28 and the 24 GiB one of N = 30 ("cache-partitioned seen-set probing, proven on the synthetic spec first")
cd /root/repo; D=gpurun_out/r05a; mkdir -p $D
timeout 120 tla_rust_amd/_build/probe_locality 8 28 > $D/probe_locality_8GiB.jsonl 2>&1; tail -12 $D/probe_locality_8GiB.jsonl | cut -c1-300
timeout 180 tla_rust_amd/_build/probe_locality 24 28 > $D/probe_locality_24GiB.jsonl 2>&1; tail -12 $D/probe_locality_24GiB.jsonl | cut -c1-300
