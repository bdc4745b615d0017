for c in 9,20,23 7,17,23 8,19,23 6,16,23 9,17,23 5,13,23 9,19,23 10,20,23 7,20,23 9,16,23; do
  timeout -k 10 120 python bench.py --steps 40 --cuts $c --no-api --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$c', d['ms_per_step'])"
done
