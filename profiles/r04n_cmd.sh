# round 4, call n (host cores of the GPU box only): deeper oracle goldens for config 5 — SI 4 x 3 level 11, and under SYMMETRY to 13 levels
cd /root/repo; D=gpurun_out/r04n; mkdir -p $D
make -s -C oracle >/dev/null 2>&1
( time timeout 900 oracle/_build/oracle_mc ssi 4 3 127 0 --threads 64 --levels 11 --levels-out > $D/oracle_ssi4x3_l11.txt ) 2> $D/oracle_ssi4x3_l11.time; cut -c1-600 $D/oracle_ssi4x3_l11.txt; tail -3 $D/oracle_ssi4x3_l11.time
( time timeout 1200 oracle/_build/oracle_mc ssi 4 3 127 0 0 3 --threads 64 --levels 13 --levels-out > $D/oracle_ssi4x3_sym_l13.txt ) 2> $D/oracle_ssi4x3_sym_l13.time; cut -c1-600 $D/oracle_ssi4x3_sym_l13.txt; tail -3 $D/oracle_ssi4x3_sym_l13.time
free -g | head -2
