#!/usr/bin/env python3
"""Build libst_amd.so for gfx950 with hipcc (cross-compiles without a GPU).

    python style-transfer-pytorch_amd/build.py [--force] [--save-temps] [--experiments]

Each csrc/*.hip is compiled to an object in parallel and linked into lib/libst_amd.so (in-tree, so
the library travels with the repo snapshot to the GPU box).  Rebuilds only what changed.

Two flavours.  The DEFAULT library holds the hot path and nothing else.  --experiments (-DST_EXPERIMENTS, objects in
build_exp/) adds the code no default path executes - the persistent Newton-Schulz chain kernel (st_nschain.hip), the Winograd
convolution (st_conv_wino.hip), the TV hazard's reproducer kernels, measurement-only kernels - and lets every ST_* switch be set
from the environment (csrc/st_common.h "build flavours").  st_has_experiments() tells which one is loaded; the tests of the
experiment code skip on a default library.
"""
import argparse
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, 'csrc')
OBJ = os.path.join(ROOT, 'build')
LIB_DIR = os.path.join(ROOT, 'lib')
LIB = os.path.join(LIB_DIR, 'libst_amd.so')
FLAVOUR = os.path.join(LIB_DIR, 'flavour.txt')      # which flavour libst_amd.so was linked from
EXPERIMENT_SOURCES = ('st_nschain.hip', 'st_conv_wino.hip')
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
         '-Wno-unused-result', '-Wno-unused-value', '-Wno-cuda-compat', '-DST_AMD_BUILD',
         # SimplifyCFG's common-code sinking merges stores to DIFFERENT constant slots of a register array
         # (sibling branches of the unrolled staging code) into one store with a selected index; the array
         # then lives in scratch memory with a vmcnt wait per element (measured: -10 % on the conv kernels)
         '-mllvm', '-simplifycfg-sink-common=false',
         '-Rpass-analysis=kernel-resource-usage']


def hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libst_amd.so cannot be built')
    return exe


def newest_header_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(os.path.dirname(ROOT), 'include')):
        for f in os.listdir(d):
            if f.endswith('.h'):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


# Hazard guard (round 5, profiles/r05_tv_hazard.md): a packed-FP32 instruction whose SECOND source takes its low lane from the
# high half of the register pair (`v_pk_add_f32 ... op_sel:[0,1] ...`, formed by the SLP vectoriser from two scalar
# subtractions of one value) returned `src0 - 0` in its low lane for lanes 48 - 63 on MI355X under co-residency - the flaky TV
# term of round 4.  No kernel of the library may contain one; the reproducer variants of tv_interior_kernel are the exception.
HAZARD = re.compile(r'v_pk_(add|mul|fma)_f32\s.*op_sel:\[[01],1(,[01])?\]')       # (two- and three-operand forms)
HAZARD_ALLOWED = ('tv_interior_kernelILi0E', 'tv_interior_kernelILi3E')


def hazard_guard(obj, has_kernels=True):
    objdump = os.path.join(os.path.dirname(os.path.realpath(hipcc())), '..', 'lib', 'llvm', 'bin', 'llvm-objdump')
    if not os.path.exists(objdump):
        objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    if not os.path.exists(objdump):
        # (advisor, round 5: the guard is the only thing that keeps the instruction out - never skip it silently)
        return ['llvm-objdump not found: the hazard guard cannot inspect %s' % os.path.basename(obj)]
    bad = []
    images = 0
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([objdump, '--offloading', os.path.basename(local)], cwd=tmp, capture_output=True, text=True)
        for name in os.listdir(tmp):
            if 'amdgcn' not in name:
                continue
            images += 1
            dis = subprocess.run([objdump, '-d', os.path.join(tmp, name)], capture_output=True, text=True).stdout
            kernel = '?'
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(\S+)>:', line)
                if m:
                    kernel = m.group(1)
                elif HAZARD.search(line) and not any(a in kernel for a in HAZARD_ALLOWED):
                    bad.append(f'{kernel}: {line.strip().split("//")[0].strip()}')
    if images == 0 and has_kernels:
        bad.append('no amdgcn code object could be extracted from %s: the hazard guard saw nothing' % os.path.basename(obj))
    return bad


# resource guard exceptions: the operator-level Winograd prototype keeps 16 accumulators (256 AGPRs) and spills two loop-invariant
# index registers (8 bytes of scratch, touched once before and once after the K loop) - not a kernel of the hot path
SPILL_ALLOWED = ('wino_conv_kernel', 'conv_fat_kernel')      # (both: a few loop-invariant registers spilled in the prologue, reloaded in the epilogue; no scratch access inside the K loop)


def compile_one(src, obj, extra):
    cmd = [hipcc(), *FLAGS, *extra, '-c', src, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    log, rc = r.stdout + r.stderr, r.returncode
    if rc == 0:
        hz = hazard_guard(obj, has_kernels='__global__' in open(src).read())
        if hz:
            rc = 1
            log += '\nhazard guard: packed-FP32 instructions with a cross-half op_sel on the second source in %s:\n  %s' % (
                src, '\n  '.join(hz[:10]))
            os.remove(obj)
    # resource guard: no kernel may use scratch memory or spill registers
    keep, bad = [], []
    function = ''
    after_remark = False
    for line in log.splitlines():
        if after_remark and re.match(r'^\s*\d*\s*\|', line):       # the source line / caret the compiler prints under a remark
            continue
        after_remark = False
        if 'kernel-resource-usage' in line or line.strip().startswith('remark:'):
            after_remark = True
            if 'Function Name:' in line:
                function = line.split('Function Name:')[1].split()[0]
            for key in ('ScratchSize [bytes/lane]:', 'VGPRs Spill:', 'SGPRs Spill:'):
                if key in line and int(line.split(key)[1].split()[0]) != 0 and key != 'SGPRs Spill:' and \
                        not any(a in function for a in SPILL_ALLOWED):
                    bad.append(line.strip())
            continue
        keep.append(line)
    log = '\n'.join(keep)
    if bad and rc == 0:
        rc = 1
        log += '\nresource guard: kernels in %s use scratch / spill VGPRs:\n  %s' % (src, '\n  '.join(bad))
        if os.path.exists(obj):
            os.remove(obj)
    return src, rc, log


def build(force=False, save_temps=False, verbose=True, experiments=False):
    obj_dir = OBJ + ('_exp' if experiments else '')
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith('.hip') and (experiments or f not in EXPERIMENT_SOURCES))
    hdr = newest_header_mtime()
    jobs, objs = [], []
    flavour = 'experiments' if experiments else 'default'
    linked = open(FLAVOUR).read().strip() if os.path.exists(FLAVOUR) else ''
    for f in sources:
        src, obj = os.path.join(CSRC, f), os.path.join(obj_dir, f[:-4] + '.o')
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr)
        if stale:
            extra = (['-save-temps=obj'] if save_temps else []) + (['-DST_EXPERIMENTS'] if experiments else [])
            jobs.append((src, obj, extra))
    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for src, rc, log in ex.map(lambda j: compile_one(*j), jobs):
                if verbose and log.strip():
                    print(log, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f'hipcc failed on {src}')
                if verbose:
                    print(f'compiled {os.path.basename(src)}')
    if jobs or not os.path.exists(LIB) or linked != flavour:
        cmd = [hipcc(), '-shared', '-fPIC', f'--offload-arch={ARCH}', *objs, '-o', LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n' + r.stdout + r.stderr)
        with open(FLAVOUR, 'w') as f:
            f.write(flavour + '\n')
        if verbose:
            print(f'linked {LIB} ({flavour})')
    return LIB


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--force', action='store_true')
    ap.add_argument('--save-temps', action='store_true')
    ap.add_argument('--experiments', action='store_true', help='also build the code no default path executes (see the docstring)')
    a = ap.parse_args()
    build(a.force, a.save_temps, experiments=a.experiments)
