"""Ahead-of-time build of the gfx950 kernel library (no torch cpp_extension, no hipify pass).

`python -m layoutdetr_amd.build` compiles every source under csrc/ with hipcc for gfx950 and links
`layoutdetr_amd/lib/libldetr_hip.so`, a C-ABI shared library (see include/ldetr_hip.h).
The reference builds its plugins lazily at first use (torch_utils/custom_ops.py:62-158); here the
library is built in-tree so that it travels with the source tree and is visibly loaded.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIBNAME = 'libldetr_hip.so'
ARCH = 'gfx950'

SOURCES = ['ldetr_core.cpp', 'bias_act.hip', 'upfirdn2d.hip', 'gemm_conv.hip', 'attention.hip', 'layernorm.hip',
           'misc_ops.hip', 'optim.hip', 'lsap.hip', 'xent.hip', 'resample.hip', 'layout_loss.hip', 'demod.hip', 'wgrad_smallc.hip', 'stem_conv.hip', 'box_ops.hip', 'ffn_fused.hip', 'conv_c32.hip', 'mha_small.hip', 'p3_engine.hip', 'bmm_strided.hip']
HEADERS = ['ldetr_common.hpp', os.path.join('..', '..', 'include', 'ldetr_hip.h')]

# the per-block tracer of the tiled kernel (tools/trace_tiles.py) is a development build: LDETR_TILE_TRACE=1 python -m layoutdetr_amd.build --force
# (the production kernel carries explicit sched_barrier(0) fences where the tracer's stamps used to sit)
FLAGS = ([] if os.environ.get('LDETR_TILE_TRACE') else ['-DLDETR_TILE_TRACE=0']) + (['-DP3_TRACE'] if os.environ.get('LDETR_P3_TRACE') else []) + ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-munsafe-fp-atomics', '-fno-gpu-rdc',
         '-Wno-unused-result', '-Rpass-analysis=kernel-resource-usage']
# kernels whose register budget is the design: any scratch (spill / stack object) is a build error, not a silent 10x slowdown
# (an erf in the engine's epilogue once cost 320 bytes of scratch per lane and every 128x128 GEMM ran 14x slower)
NO_SCRATCH = ('wgrad_c32_3x3_kernel', 'gemm_f32_kernel', 'gemm_small_kernel', 'attn_fwd_kernel', 'attn_fwd_wide_kernel', 'attn_bwd_kernel', 'ffn_fwd_kernel', 'ffn_bwd_kernel', 'conv3x3_c32_kernel', 'mha_small_fwd_kernel', 'mha_cross_fwd_kernel', 'mha_small_bwd_kernel', 'mha_cross_bwd_kernel', 'wgrad_multi_kernel', 'ln_fwd_kernel', 'ln_bwd_kernel', 'p3_nt_kernel', 'p3_tn_kernel', 'p3_c3_kernel', 'p3_bwd_pair')


def _hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found: the gfx950 kernel library cannot be built')


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _kernel_resources(remarks):
    """Parse clang's -Rpass-analysis=kernel-resource-usage remarks -> {mangled kernel name: {field: value}}."""
    out, cur = {}, None
    for line in remarks.splitlines():
        if 'remark:' not in line:
            continue
        body = line.split('remark:', 1)[1].split('[-Rpass-analysis', 1)[0].strip()
        if body.startswith('Function Name:'):
            cur = body.split(':', 1)[1].strip()
            out[cur] = {}
        elif cur is not None and ':' in body:
            k, v = body.rsplit(':', 1)
            out[cur][k.strip()] = v.strip()
    return out


def source_digest():
    """sha256 over the kernel sources and headers: identifies the build a measurement (profiles/pmc_traffic.json) belongs to."""
    return _digest(sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp', '.hpp')) and not f.startswith('.'))
                   + [os.path.normpath(os.path.join(CSRC, HEADERS[1]))])


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src + '.o')
        stamp = obj + '.sha'
        dig = _digest([sp] + hdrs)
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, obj, stamp, dig))

    def compile_one(job):
        sp, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + ['-x', 'hip', '-c', sp, '-o', obj]
        if verbose:
            print('[ldetr build]', os.path.basename(sp), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}')
        res = _kernel_resources(r.stderr)
        with open(obj + '.resources.txt', 'w') as f:
            for name, d in res.items():
                f.write(f"{name} vgprs={d.get('VGPRs', '?')} agprs={d.get('AGPRs', '?')} scratch={d.get('ScratchSize [bytes/lane]', '?')} "
                        f"occupancy={d.get('Occupancy [waves/SIMD]', '?')} lds={d.get('LDS Size [bytes/block]', '?')}\n")
        bad = [n for n, d in res.items() if any(k in n for k in NO_SCRATCH) and d.get('ScratchSize [bytes/lane]', '0') != '0']
        if bad:
            raise RuntimeError(f'{os.path.basename(sp)}: scratch memory in register-budgeted kernels: ' + ', '.join(bad[:4]))
        with open(stamp, 'w') as f:
            f.write(dig)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    out = lib_path()
    if jobs or not os.path.exists(out):
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-fno-gpu-rdc', '-o', out] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print('[ldetr build] linked', out, flush=True)
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
