#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses sit: per kernel of a hipcc -S file, the static scratch size and the number of scratch_load /
scratch_store instructions in basic blocks that also hold matrix instructions (the node loop's hot blocks) against all other blocks (set-up,
rare branches, exits).  A spill in a cold block costs nothing per node; one in a hot block goes to HBM on every node.

    python tools/isa_scratch.py file.s [name-pattern ...]
"""
import re
import sys


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)]
    for st in starts:
        name = lines[st].split(':')[0]
        if pats and not all(p in name for p in pats):
            continue
        end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
        blocks, cur = [], dict(label='entry', n=0, scratch=0, mfma=0)
        for l in lines[st + 1:end]:
            m = re.match(r'^(\.LBB\S+):', l)
            if m:
                blocks.append(cur)
                cur = dict(label=m.group(1), n=0, scratch=0, mfma=0)
            elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
                cur['n'] += 1
                cur['scratch'] += 'scratch_' in l
                cur['mfma'] += 'v_mfma' in l
        blocks.append(cur)
        hot = [b for b in blocks if b['mfma'] >= 8]
        cold = [b for b in blocks if b['mfma'] < 8]
        size = next((int(m.group(1)) for l in lines[end:end + 400] for m in [re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', l)] if m), -1)
        print(f"{name[:110]}\n   scratch bytes/lane {size}; hot blocks: {sum(b['n'] for b in hot)} instr, {sum(b['mfma'] for b in hot)} mfma, "
              f"{sum(b['scratch'] for b in hot)} scratch ops; other blocks: {sum(b['n'] for b in cold)} instr, {sum(b['scratch'] for b in cold)} scratch ops")


if __name__ == '__main__':
    main()
