"""Per-kernel resource use + the skeleton of the main loop's memory operations / waits / barriers."""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
verbose = len(sys.argv) > 2
for m in re.finditer(r'\n(_ZN3mww\S+):[^\n]*\n(.*?)\.amdhsa_kernel \1\n(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body, meta = m.group(1), m.group(2), m.group(3)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', meta) or [None, '?'])[1]
    print('==', dem.replace('mww::', ''), '| vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'lds', g('group_segment_fixed_size'), 'scratch', g('private_segment_fixed_size'))
    lines = body.split('\n')
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
