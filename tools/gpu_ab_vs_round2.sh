# GPU session 2 of round 3: A/B against the round-2 tree on the same box, interrupt vs polling completion, then the
# full round-3 session (tools/gpu_round3.sh).
set -x
O=gpurun_out/r3b
mkdir -p $O
R=$PWD
for i in 1 2; do
  (cd variants/r02_tree && timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 30 > $R/$O/ab_r02_steps200_$i.json 2>/dev/null)
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 30 > $O/ab_r03_steps200_$i.json 2>/dev/null
  (cd variants/r02_tree && timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $R/$O/ab_r02_steps20_$i.json 2>/dev/null)
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/ab_r03_steps20_$i.json 2>/dev/null
  HSA_ENABLE_INTERRUPT=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > $O/ab_r03_steps20_irq_$i.json 2>/dev/null
done
for w in "synthetic50x20 8192" "mixed 32768"; do set -- $w
  (cd variants/r02_tree && timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --workload $1 --batch $2 > $R/$O/ab_r02_$1.json 2>/dev/null)
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 200 --workload $1 --batch $2 > $O/ab_r03_$1.json 2>/dev/null
  timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --workload $1 --batch $2 > $O/ab_r03_$1_steps20.json 2>/dev/null
done
