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
        # the node loop: the SMALLEST backward-branch span that holds the kernel's matrix instructions (an enclosing loop -- the flush segments
        # of the fp16 x 2 backward kernels, which also hold the table fill and the end-of-kernel combine -- is not it); kernels without matrix
        # instructions: the largest span
        n_mfma = sum('v_mfma' in l for l in body)
        spans = []
        for i, l in enumerate(body):
            m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\S+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                lo = labels[m.group(1)]
                spans.append((i - lo, lo, i, sum('v_mfma' in b for b in body[lo:i + 1])))
        full = [sp for sp in spans if n_mfma and sp[3] == n_mfma]
        span = min(full)[:3] if full else (max(spans)[:3] if spans else (0, 0, 0))
        loop = [l for l in body[span[1]:span[2] + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        in_loop = sum('scratch_' in l for l in loop)
        total = sum(b['scratch'] for b in blocks)
        with_mfma = sum(b['scratch'] for b in blocks if b['mfma'] >= 8)
        print(f"{name[:110]}\n   node loop: {len(loop)} instr, {sum('v_mfma' in l for l in loop)} mfma, {in_loop} scratch ops ({with_mfma} in blocks with matrix "
              f"instructions); outside the loop: {total - in_loop} scratch ops")


if __name__ == '__main__':
    main()
