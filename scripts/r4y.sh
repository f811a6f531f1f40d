mkdir -p gpurun_out/r4y
BO_SELECT_PROFILE=1 timeout -k 10 1100 python bench.py --fast --games 65536 --select-sweep --roofline-steps 24 > gpurun_out/r4y/sweep65536.log 2> gpurun_out/r4y/sweep65536.err; echo "rc=$?"; grep "sweep\|select profile\|Error\|error" gpurun_out/r4y/sweep65536.err | tail -12; tail -c 900 gpurun_out/r4y/sweep65536.log
