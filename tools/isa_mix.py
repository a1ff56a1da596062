#!/usr/bin/env python3
"""Instruction mix of the kernels in a hipcc --save-temps .s file whose mangled name contains every given pattern.

    python tools/isa_mix.py file.s node_bwd_x3_kernelILi1ELi2ELi2ELi32ELi0E [--loop]
With --loop only the largest backward-branch loop body (the node loop) is counted.
"""
import collections
import re
import sys


def main():
    path, pats = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith('--')]
    loop_only = '--loop' in sys.argv
    lines = open(path).read().split('\n')
    starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l)]
    for si, st in enumerate(starts):
        name = lines[st].split(':')[0]
        if not all(p in name for p in pats):
            continue
        end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
        body = lines[st + 1:end]
        if loop_only:
            labels = {l.split(':')[0]: i for i, l in enumerate(body) if re.match(r'^\.LBB\S+:', l)}
            best = (0, 0, len(body))
            for i, l in enumerate(body):
                m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\S+)', l)
                if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[0]:
                    best = (i - labels[m.group(1)], labels[m.group(1)], i)
            body = body[best[1]:best[2]]
        c = collections.Counter(l.split()[0] for l in body if l.startswith('\t') and not l.strip().startswith(('.', ';')))
        print(name[:90], 'instructions:', sum(c.values()))
        print('   ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(32)))


if __name__ == '__main__':
    main()
