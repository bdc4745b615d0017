# GPU box: pipeline cut positions (layer index where a new stage starts).  usage: tools/cut_sweep.sh > gpurun_out/<tag>_cuts.txt
for c in 9,20,23 9,19,23 9,17,23 8,19,23 10,20,23 9,21,23 9,22,23 10,21,23 11,21,23 9,20,22 7,17,23 6,16,23 9,16,23 12,20,23 10,17,23; do
  timeout -k 10 120 python bench.py --steps 60 --cuts $c --no-api --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'], flush=True)"
done
