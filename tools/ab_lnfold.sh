for i in 1 2 3; do
for f in "" "--no-ln-fold"; do
python bench.py --no-cpu-baseline --no-rollouts --no-profile --steps 20 --warmup 3 $f 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f', d['value'], d['ms_per_step'])"
done; done
