#!/bin/bash
# Compare the gfx950 ISA of two objects kernel by kernel (instruction text only, addresses stripped):
#   tools/isa_diff.sh old.o new.o [kernel-name-fragment]
# Used to prove that a refactoring left a hot kernel's code untouched.
set -u
B=/opt/rocm/lib/llvm/bin
for tag in a b; do
  f=$1; [ $tag = b ] && f=$2
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $f /tmp/isa_$tag.fb
  $B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=/tmp/isa_$tag.fb --output=/tmp/isa_$tag.co
  $B/llvm-objdump -d --no-show-raw-insn /tmp/isa_$tag.co | sed -E 's/^\s*([a-z_0-9]+.*)\/\/ [0-9A-F]+:.*$/\1/; s/\s+$//' > /tmp/isa_$tag.txt
done
python3 - "${3:-}" <<'PY'
import re, sys
frag = sys.argv[1]
def kernels(path):
    out, cur = {}, None
    for ln in open(path):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', ln.strip())
        if m:
            cur = re.sub(r'8ZeroFilljj$', '8ZeroFill', m.group(1)); out[cur] = []
        elif cur is not None and ln.strip():
            out[cur].append(re.sub(r'<[^>]*>', '<L>', ln.strip()))
    return out
A, B = kernels('/tmp/isa_a.txt'), kernels('/tmp/isa_b.txt')
for k in sorted(set(A) | set(B)):
    if frag and frag not in k:
        continue
    a, b = A.get(k), B.get(k)
    if a is None or b is None:
        print(('ONLY-OLD ' if b is None else 'ONLY-NEW ') + k[:150]); continue
    print(('SAME     ' if a == b else f'DIFF {len(a)}->{len(b)} ') + k[:150])
PY
