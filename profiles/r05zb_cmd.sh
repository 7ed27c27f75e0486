# round 5, GPU call zb — THE LAST TREE (r05z + the host-side changes of the last hours: record parameters, `with v = r`, unrolled small
# intervals, CHECK_DEADLOCK, the Paxos model; device code unchanged since r05y): the whole GPU suite as the driver runs it, and smoke()
cd /root/repo; D=$PWD/gpurun_out/r05zb; mkdir -p $D
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > $D/pytest_gpu_full.log 2>&1; grep -E 'passed|failed|error|s call' $D/pytest_gpu_full.log | tail -8; grep -E "^(FAILED|ERROR)" $D/pytest_gpu_full.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $D/smoke.log 2>&1; tail -n 1 $D/smoke.log
