import sys, json, subprocess, os
# usage: ab.py lib1 lib2 ... : runs bench (C5 20k pairs) with each library, interleaved rounds
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, ALN_LIB=l)
        out = subprocess.run([sys.executable, "bench.py", "--pairs", "20000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-single-pair"], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        d = json.loads(out)
        res[l].append((d["value"], d["roofline"]["kernel_ms"], d["roofline"]["traceback_ms"]))
for l in libs:
    print(l, res[l])
