"""Developer tool: run bench.py with the given extra flags and print a short summary."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline'] + sys.argv[1:],
                     capture_output=True, text=True)
lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
if not lines:
    print(out.stdout[-2000:], out.stderr[-4000:])
    sys.exit(1)
d = json.loads(lines[-1])
print(f"{d['value']:.0f} Mpix/s  {d['ms_per_step']:.2f} ms/step  roofline frac {d['roofline']['frac']:.3f}")
print({k: (round(v['ms_per_step'], 2), round(v['avg_us'], 1)) for k, v in d['kernels'].items()})
