"""Basic blocks of one kernel in a `hipcc -S` listing that hold the erf terms (at least MIN v_rcp_f32): instructions, VALU, packed,
moves, scratch accesses, wait states.  python tools/loop_bodies.py listing.s kernel_symbol [min_rcp]"""
import re, sys
def analyse(path, kname, min_rcp=4):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith(kname + ':') or l.startswith(kname + ': ')][0]
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    blocks, cur = [], ('entry', [])
    for l in lines[start:end]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur); cur = (m.group(1), [])
        elif l.startswith('\t') and not l.strip().startswith(';') and not l.strip().startswith('.'):
            cur[1].append(l.strip())
    blocks.append(cur)
    for name, ins in blocks:
        c = lambda pre: sum(1 for i in ins if i.startswith(pre))
        if c('v_rcp_f32') >= min_rcp:
            print(f"{name:12s} instr {len(ins):4d} valu {c('v_'):4d} rcp {c('v_rcp_f32'):3d} pk {c('v_pk_'):3d} v_mov {c('v_mov_b'):3d} scratch {c('scratch_')} s_nop {c('s_nop')} ds {c('ds_')}")
if __name__ == '__main__':
    analyse(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4)
