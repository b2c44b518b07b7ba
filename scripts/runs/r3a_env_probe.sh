set -e
mkdir -p gpurun_out/r3a
B="--steps 100 --warmup 8 --no-cpu-baseline --no-secondary --no-roofline"
./scripts/boundary_probe 300 > gpurun_out/r3a/probe_unset.txt 2>&1
HIP_FORCE_DEV_KERNARG=1 ./scripts/boundary_probe 300 > gpurun_out/r3a/probe_dev1.txt 2>&1
HIP_FORCE_DEV_KERNARG=0 ./scripts/boundary_probe 300 > gpurun_out/r3a/probe_dev0.txt 2>&1
for i in 1 2; do
python3 bench.py $B > gpurun_out/r3a/bench_base_$i.json 2> gpurun_out/r3a/bench_base_$i.err
HIP_FORCE_DEV_KERNARG=1 python3 bench.py $B > gpurun_out/r3a/bench_dev1_$i.json 2> gpurun_out/r3a/bench_dev1_$i.err
done
HIP_FORCE_DEV_KERNARG=0 python3 bench.py $B > gpurun_out/r3a/bench_dev0.json 2> gpurun_out/r3a/bench_dev0.err
HIP_FORCE_DEV_KERNARG=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 python3 bench.py $B > gpurun_out/r3a/bench_dev1_pkt.json 2> gpurun_out/r3a/bench_dev1_pkt.err
grep -h ms_per_step gpurun_out/r3a/*.json | cut -c1-200
