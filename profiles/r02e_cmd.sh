# round 2, GPU call e: new tests (progress lines, negative control, overflow, checkpoint binding) + FETCH_SIZE / WRITE_SIZE calibration
cd /root/repo; mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -x -q -k "progress or negative_control or overflow or compiled_program_is_bound or corrupt_level or budget_stops or cut_at_its_depth" > gpurun_out/r02e/pytest_new.log 2>&1; tail -5 gpurun_out/r02e/pytest_new.log
cd /tmp && export TMPDIR=/tmp
C=/root/repo/tla_rust_amd/_build/calib_fetch
$C > /root/repo/gpurun_out/r02e/calib_plain.jsonl 2>&1
for set in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set --output-format csv -d /tmp/calib_$name -- $C > /tmp/calib_$name.log 2>&1
  cp /tmp/calib_$name/*/*_counter_collection.csv /root/repo/gpurun_out/r02e/calib_$name.csv 2>/dev/null
done
cd /root/repo
python profiles/summarize_pmc.py gpurun_out/r02e/calib_pmc.json gpurun_out/r02e/calib_*.csv > gpurun_out/r02e/calib_summary.txt 2>&1
cat gpurun_out/r02e/calib_plain.jsonl; cat gpurun_out/r02e/calib_summary.txt | cut -c1-300
