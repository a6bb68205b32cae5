for args in "--in-flight 1" "--in-flight 2" "--in-flight 2 --free-overlap" "--in-flight 3" "--in-flight 3 --free-overlap" "--size 2048 --batch 32" "--size 4096 --batch 8" "--size 16384 --in-flight 1" "--size 4096 --in-flight 2" "--mode3 --in-flight 1" "--size 2048 --in-flight 8"; do
  echo "== $args"; timeout -k 10 200 python bench.py --steps 50 --warmup 5 --no-cpu --no-parity $args 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['other_kernels_ms'])"
done
