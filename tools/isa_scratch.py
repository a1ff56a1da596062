#!/usr/bin/env python3
"""Where a kernel's scratch (spill) accesses sit: per kernel of a hipcc -S file, the number of scratch_load / scratch_store instructions inside
the node loop (the smallest backward-branch span that holds all of the kernel's matrix instructions; blocks of rare branches that the compiler placed inside the span are told apart
by holding no matrix instruction AND being jumped over) against the rest (set-up, exits).  A spill outside the loop costs nothing per node;
one inside is a memory round trip on every node -- eleven reloads of spilled plane addresses made the order-3 gates backward 27 % slower
while every block with matrix instructions was clean.

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
        body = lines[st + 1:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r'^(\.LBB\S+):', l)] if m}
        # the node loops: every backward-branch span that holds matrix instructions and no smaller such span with as many of them (an enclosing
        # loop -- the loop over PASSES of the fp16 x 2 backward kernels, which also holds the table fill and the combine -- is not a node loop).
        # Those kernels compile their body twice: the first pass as straight-line code (the loop every launch runs: listed first), later passes
        # (rare) inside the pass loop.  Kernels without matrix instructions: the largest span.
        spans = []
        for i, l in enumerate(body):
            m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\S+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo = labels[m.group(1)]
                spans.append((i - lo, lo, i, sum('v_mfma' in b for b in body[lo:i + 1])))
        with_mfma = [sp for sp in spans if sp[3] > 0]
        loops = [sp for sp in with_mfma if not any(o is not sp and o[1] >= sp[1] and o[2] <= sp[2] and o[3] == sp[3] and o[0] < sp[0] for o in with_mfma)]
        loops = sorted(loops, key=lambda sp: sp[1]) or ([max(spans)] if spans else [(0, 0, 0, 0)])
        total = sum(b['scratch'] for b in blocks)
        print(name[:110])
        for k, span in enumerate(loops):
            loop = [l for l in body[span[1]:span[2] + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
            in_loop = sum('scratch_' in l for l in loop)
            print(f"   node loop{'' if len(loops) == 1 else f' {k + 1}/{len(loops)}'}: {len(loop)} instr, {sum('v_mfma' in l for l in loop)} mfma, {in_loop} scratch ops; "
                  f"kernel total: {total} scratch ops")


if __name__ == '__main__':
    main()
