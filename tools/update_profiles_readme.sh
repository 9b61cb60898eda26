#!/bin/bash
# Developer tool: (re)place the generated section of round <tag> in profiles/README.md, in front of the earlier rounds.
#   bash tools/update_profiles_readme.sh r05
tag=${1:-r05}
python3 - "$tag" <<'PY'
import re, subprocess, sys
tag = sys.argv[1]
n = int(tag[1:])
sec = subprocess.run([sys.executable, 'tools/make_profiles_readme.py', tag], capture_output=True, text=True, check=True).stdout.strip() + '\n\n'
p = 'profiles/README.md'
s = open(p).read()
s = re.sub(rf'Round {n}: the `{tag}_\*` files.*?(?=Round {n - 1}: )', '', s, flags=re.S)
i = s.index(f'Round {n - 1}: ')
open(p, 'w').write(s[:i] + sec + s[i:])
PY
