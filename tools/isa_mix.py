#!/usr/bin/env python
'''Static instruction mix of the library's gfx950 kernels (build container, no GPU needed): unbundles the device
code object of every csrc/build/*.o, disassembles it with llvm-objdump and prints, per kernel whose name contains
one of the given substrings, registers / scratch and the instruction classes of the whole kernel and of its
largest loops.  This is how round 6 found the 656 ds_bpermute_b32 of the head kernels' epilogues.
    python tools/isa_mix.py lstm_bwd_rs_kernelILi16 sep_pit_fwd anchor_fwd ...'''
import os, re, struct, subprocess, sys, tempfile
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, 'danet-tensorflow_amd', 'csrc', 'build')
LLVM = '/opt/rocm/lib/llvm/bin'


def code_object(obj, out):
    b = open(obj, 'rb').read()
    i = b.find(b'__CLANG_OFFLOAD_BUNDLE__')
    if i < 0:
        return False
    n = struct.unpack_from('<Q', b, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, sz, tl = struct.unpack_from('<QQQ', b, off); off += 24
        t = b[off:off + tl].decode(); off += tl
        if 'gfx950' in t and sz:
            open(out, 'wb').write(b[i + o:i + o + sz])
            return True
    return False


def classes(ins):
    c = Counter()
    for _, op, _ in ins:
        c[op.split('_')[0]] += 1
        for key, pred in (('mfma', op.startswith('v_mfma')), ('exp', op.startswith('v_exp')),
                          ('div/rcp', op.startswith('v_div') or op.startswith('v_rcp')),
                          ('bpermute', 'bpermute' in op), ('dpp', False), ('scratch', op.startswith('scratch')),
                          ('vmem_ld', op.startswith(('global_load', 'buffer_load'))),
                          ('vmem_st', op.startswith(('global_store', 'buffer_store'))),
                          ('waitcnt', op == 's_waitcnt'), ('barrier', op == 's_barrier'),
                          ('cndmask', op.startswith('v_cndmask')), ('mov', op.startswith(('v_mov', 'v_accvgpr')))):
            if pred:
                c[key] += 1
    return dict(c)


def main():
    pats = sys.argv[1:] or ['lstm_bwd_rs_kernelILi16ELi1', 'lstm_fwd_fx_kernelILi8ELi3']
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(BUILD)):
            if not f.endswith('.o'):
                continue
            co = os.path.join(tmp, f + '.co')
            if not code_object(os.path.join(BUILD, f), co):
                continue
            notes = subprocess.run([LLVM + '/llvm-readelf', '--notes', co], capture_output=True, text=True).stdout
            meta = {}
            for blk in notes.split('- .agpr_count')[1:]:
                nm = re.search(r'\.name:\s+(\S+)', blk)
                if nm:
                    meta[nm.group(1)] = {k: int(v) for k, v in re.findall(r'\.(vgpr_count|sgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill_count):\s+(\d+)', '.agpr_count' + blk)}
            txt = subprocess.run([LLVM + '/llvm-objdump', '-d', '--mcpu=gfx950', co], capture_output=True, text=True).stdout.split('\n')
            starts = [(i, l) for i, l in enumerate(txt) if re.match(r'^[0-9a-f]{16} <.*>:$', l)]
            for k, (i, l) in enumerate(starts):
                name = l.split('<')[1][:-2]
                if not any(p in name for p in pats):
                    continue
                end = starts[k + 1][0] if k + 1 < len(starts) else len(txt)
                ins = []
                for ln in txt[i + 1:end]:
                    m = re.match(r'^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-F]+):', ln)
                    if m:
                        ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
                addr = {a: idx for idx, (a, _, _) in enumerate(ins)}
                loops = []
                for idx, (a, op, args) in enumerate(ins):
                    if op.startswith('s_cbranch') or op == 's_branch':
                        m = re.match(r'(\d+)', args)
                        if m:
                            d = int(m.group(1))
                            d -= 65536 if d >= 32768 else 0
                            tgt = a + 4 + d * 4
                            if tgt in addr and addr[tgt] < idx:
                                loops.append((addr[tgt], idx))
                print('%s  (%s)' % (name[:90], f))
                print('   ', meta.get(name, {}))
                print('    whole kernel %d: %s' % (len(ins), classes(ins)))
                seen = set()
                for s, e in sorted(loops, key=lambda x: -(x[1] - x[0])):
                    if any(abs(s - s2) < 8 and abs(e - e2) < 40 for s2, e2 in seen):
                        continue
                    seen.add((s, e))
                    if len(seen) > 3:
                        break
                    print('    loop [%d..%d] %d: %s' % (s, e, e - s + 1, classes(ins[s:e + 1])))


if __name__ == '__main__':
    main()
