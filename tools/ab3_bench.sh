for i in 1 2; do for v in build/libA.so build/libB.so build/libC.so; do cp $v yaik_amd/libyaik_hip.so; echo "== $v"; timeout -k 10 200 python bench.py --steps 100 --warmup 5 --no-cpu --no-parity 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done; done
cp build/libA.so yaik_amd/libyaik_hip.so
