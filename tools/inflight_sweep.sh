for f in 2 3 4 2 3; do
  timeout -k 10 200 python bench.py --in-flight $f --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('inflight', $f, d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
