'''
Builds libdanet_hip.so (gfx950 only) in-tree with hipcc.

    python danet-tensorflow_amd/_build.py [--force]

One object per .hip/.cpp under csrc/, compiled in parallel, linked into
csrc/libdanet_hip.so.  Objects are rebuilt only when their source (or a shared
header) is newer.  No torch headers are involved: the library is a plain C ABI
(include/danet_hip.h).
'''
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
BUILD = os.path.join(CSRC, 'build')
LIB = os.path.join(CSRC, 'libdanet_hip.so')
ARCH = 'gfx950'
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -fvisibility=hidden: only what include/danet_hip.h declares (inside its visibility pragma) is exported
FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden', '-I' + INCLUDE,
         '-I' + CSRC, '-Wno-unused-result']
# extra -D switches for A/B builds, e.g. DANET_BUILD_DEFS='-DBK=32' (use --force)
FLAGS += os.environ.get('DANET_BUILD_DEFS', '').split()


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers_mtime():
    m = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith('.h'):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def _compile(src, force, hdr_m, bdir=BUILD, defs=()):
    obj = os.path.join(bdir, os.path.splitext(src)[0] + '.o')
    sp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) > max(os.path.getmtime(sp), hdr_m)):
        return obj, False
    cmd = [HIPCC] + FLAGS + list(defs) + ['-c', sp, '-o', obj]
    if src.endswith('.hip'):
        cmd[1:1] = ['-x', 'hip']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj, True


def _link(objs, out):
    cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC',
           '-Wl,--version-script=' + os.path.join(CSRC, 'exports.map'), '-o', out] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))


def build(force=False, verbose=True):
    os.makedirs(BUILD, exist_ok=True)
    hdr_m = _headers_mtime()
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_m), srcs))
    objs = [o for o, _ in res]
    rebuilt = any(r for _, r in res)
    if rebuilt or not os.path.exists(LIB):
        _link(objs, LIB)
    if verbose:
        print('libdanet_hip.so: %s (%d objects, %s)' % (
            LIB, len(objs), 'rebuilt' if rebuilt else 'up to date'))
    return LIB


def build_variant(name, defs):
    '''A/B variant csrc/libdanet_hip_<name>.so compiled with extra -D switches (e.g.
    name='accmath', defs=['-DDANET_LSTM_ACCURATE_MATH']; name='trace',
    defs=['-DDANET_LSTM_TRACE']: the persistent LSTM kernels time-stamp every step,
    tools/trace_lstm.py); loaded instead of the product library when DANET_LIB_PATH points at it
    (tests / tools only).  Objects under csrc/build_<name>/.'''
    out = os.path.join(CSRC, 'libdanet_hip_%s.so' % name)
    bdir = os.path.join(CSRC, 'build_' + name)
    os.makedirs(bdir, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, True, 0.0, bdir, defs), srcs))
    _link([o for o, _ in res], out)
    return out


def build_trace():
    return build_variant('trace', ['-DDANET_LSTM_TRACE'])


if __name__ == '__main__':
    if '--trace' in sys.argv:
        print(build_trace())
    elif '--variant' in sys.argv:
        i = sys.argv.index('--variant')
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:]))
    else:
        build(force='--force' in sys.argv)
