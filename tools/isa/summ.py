"""Per-kernel resource use + the skeleton of the main loop's memory operations / waits / barriers.
`reloads in loops`: scratch reloads that sit inside a loop - each waits with s_waitcnt vmcnt(0), i.e. for every load requested
before it, the next tile's prefetch included (round 6: the K = 21 wide block backward lost 1-1.5 us per launch to two of them)."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
verbose = len(sys.argv) > 2
for m in re.finditer(r'\n(_ZN3mww\S+):[^\n]*\n(.*?)\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body, meta = m.group(1), m.group(2), m.group(3)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', meta) or [None, '?'])[1]
    lines = body.split('\n')
    # a block label carries "in Loop:" / "Loop Header" when it belongs to a loop: reloads between such a label and the next label
    in_loop, reloads = False, 0
    for l in lines:
        if re.match(r'\.LBB\d+_\d+:', l):
            in_loop = 'Loop' in l
        elif in_loop and 'scratch_load' in l:
            reloads += 1
    print('==', dem.replace('mww::', ''), '| vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'),
          *(['reloads in loops', reloads] if g('private_segment_fixed_size') not in ('0', '?') else []))
    # main loop = from the first "Loop Header: Depth=1" that contains an s_barrier to the end
    sk = []
    for i, l in enumerate(lines):
        s = l.strip()
        if re.match(r'(global_load|global_store|buffer_load|buffer_store|global_atomic|s_barrier)', s):
            sk.append(s.split()[0].replace('global_', 'g').replace('buffer_', 'b').replace('_dword', ''))
        elif s.startswith('s_waitcnt') and 'vmcnt' in s:
            sk.append(re.search(r'vmcnt\((\d+)\)', s).group(0))
        elif 'Loop Header: Depth=1' in l:
            sk.append('\n   [loop]')
    # compress repeats
    out, prev, n = [], None, 0
    for t in sk + [None]:
        if t == prev:
            n += 1
        else:
            if prev is not None:
                out.append(prev + ('*%d' % n if n > 1 else ''))
            prev, n = t, 1
    print('  ', ' '.join(out))
