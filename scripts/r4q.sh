mkdir -p gpurun_out/r4q
for a in "3 1 64 8 1500 eager" "3 1 64 8 1000 graph" "15 5 256 1 800 eager" "15 5 256 1 500 graph" "3 0 64 16 1000 graph" "8 2 128 3 500 graph"; do
timeout -k 10 300 python scripts/b1_stress.py $a > gpurun_out/r4q/stress_$(echo $a | tr ' ' '_').log 2>&1; tail -3 gpurun_out/r4q/stress_$(echo $a | tr ' ' '_').log
done
timeout -k 10 300 python scripts/resume_probe.py > gpurun_out/r4q/resume.log 2>&1; echo "resume probe:"; grep -v "aborted" gpurun_out/r4q/resume.log | tail -4
