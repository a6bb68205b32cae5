#!/usr/bin/env python3
"""Static VALU / SALU / memory instruction count of one kernel per source line (hipcc -gline-tables-only -save-temps).

usage: tools/isa_by_line.py <file.hip> <kernel-symbol-substring> [extra hipcc flags...]
Prints instructions per source line of the kernel body, and totals per [region] given as YK-REGION comments is not needed:
regions are line ranges passed with --regions a-b:name,...
"""
import re, subprocess, sys, tempfile, os, collections

def main():
    args = sys.argv[1:]
    regions = []
    if "--regions" in args:
        i = args.index("--regions"); spec = args[i + 1]; del args[i:i + 2]
        for part in spec.split(","):
            rng, name = part.split(":"); a, b = rng.split("-"); regions.append((int(a), int(b), name))
    src, sym, extra = args[0], args[1], args[2:]
    tmp = tempfile.mkdtemp(prefix="isa_")
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-w", "-gline-tables-only",
           "-save-temps", "-c", os.path.abspath(src), "-o", "x.o"] + extra
    subprocess.run(cmd, cwd=tmp, check=True)
    sfile = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
    lines = open(os.path.join(tmp, sfile)).read().split("\n")
    # file table
    files = {}
    inside = False
    cur = (0, 0)
    per = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
    base = os.path.basename(src)
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2))
            continue
        if re.match(r'^[A-Za-z_].*:\s*(;.*)?$', ln) and not ln.startswith("."):
            name = ln.split(":")[0]
            if sym in name: inside = True
            elif inside and name.startswith("_Z"): inside = False
        if not inside: continue
        if ".end_amdhsa_kernel" in ln or ln.strip().startswith(".Lfunc_end"): inside = False; continue
        m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
        if m:
            cur = (int(m.group(1)), int(m.group(2))); continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith("."): continue
        op = t.split()[0]
        if op.endswith(":"): continue
        kind = None
        if op.startswith("v_"): kind = "valu"
        elif op.startswith("s_"): kind = "salu"
        elif op.startswith("ds_"): kind = "lds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): kind = "vmem"
        if kind is None: continue
        f = os.path.basename(files.get(cur[0], "?"))
        key = (f, cur[1])
        kinds[key][kind] += 1
    tot = collections.Counter()
    for k, c in kinds.items():
        for kk, n in c.items(): tot[kk] += n
    print("total", dict(tot))
    if regions:
        reg = collections.defaultdict(collections.Counter)
        for (f, l), c in kinds.items():
            name = "other:" + f if f != base else None
            if f == base:
                name = "unassigned"
                for a, b, n in regions:
                    if a <= l <= b: name = n; break
            for kk, n in c.items(): reg[name][kk] += n
        for name, c in sorted(reg.items(), key=lambda x: -x[1]["valu"]):
            print(f"{name:40s} valu {c['valu']:5d} salu {c['salu']:5d} lds {c['lds']:4d} vmem {c['vmem']:4d}")
    else:
        for (f, l), c in sorted(kinds.items()):
            print(f"{f}:{l:5d} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")

if __name__ == "__main__":
    main()
