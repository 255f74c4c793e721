"""Build liblanemap_hip.so (gfx950) in-tree with hipcc.  `python -m lanemapping_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'liblanemap_hip.so')
SOURCES = ['errors.cpp', 'conv_mfma.hip', 'conv_wino44.hip', 'conv_direct.hip', 'norm_resize.hip', 'vit.hip', 'head.hip',
           'decode.hip', 'raster.hip', 'rowref.hip', 'prim.hip', 'lidar.hip', 'postproc.cpp', 'backproject.cpp', 'png_reader.cpp', 'lane_json.cpp', 'merge_lines.cpp', 'skeleton.cpp']


# integer-output kernels whose fp32 index math must match the C oracle bit for bit
EXACT_FP = {'raster.hip', 'lidar.hip', 'backproject.cpp', 'merge_lines.cpp'}
# per-source extra flags.  conv_wino44.hip: the SLP vectoriser packs the Winograd transform's adds into v_pk_add_f32 plus a dozen
# v_mov shuffles per group; packed f32 VALU beside MFMAs costs issue slots (MI355X_MICROARCH.md, filler price list)
EXTRA_FLAGS = {'conv_wino44.hip': ['-fno-slp-vectorize']}


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, 'build', src + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden',
               '-x', 'hip', '-c', os.path.join(CSRC, src), '-o', obj]
        if src in EXACT_FP:
            cmd.insert(4, '-ffp-contract=off')
        for fl in EXTRA_FLAGS.get(src, []):
            cmd.insert(4, fl)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print(f'built {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
