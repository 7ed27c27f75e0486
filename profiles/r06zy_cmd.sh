# round 6, GPU call zy: phase profile of the by-pairs kernel on generated code (the generated unit built with -DMC_PHASE_PROF): where a wavefront's time goes, and how long a level takes
cd /root/repo; D=$PWD/gpurun_out/r06zy; mkdir -p $D
timeout 600 python profiles/phase_prof_gen.py 2>$D/err.txt | tee $D/phase_profile_gen.jsonl | cut -c1-1500
grep -v amdgpu.ids $D/err.txt | tail -3
