set -x
mkdir -p gpurun_out/r4l
for f in 1 0; do
B1_F32=$f timeout -k 10 300 python scripts/b1_probe.py 15 5 256 1 > gpurun_out/r4l/b1_probe_f32_$f.log 2>&1
echo "f32=$f probe rc=$?" >> gpurun_out/r4l/b1_probe_f32_$f.log
grep -h "precision\|tower    graph\|wave \|rc=" gpurun_out/r4l/b1_probe_f32_$f.log
done
