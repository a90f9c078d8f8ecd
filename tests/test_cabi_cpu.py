"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dgr_hip.h declares; the product path fails loudly without a GPU (no CPU fallback)."""
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'dgr_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dgr_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_are_bound_and_exported():
    from deepglobalregistration_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in dgr_hip.h but not exported'
    assert b'gfx950' in lib.dgr_version()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    import numpy as np
    from deepglobalregistration_amd import ops
    from deepglobalregistration_amd.core.knn import find_knn_gpu
    with pytest.raises((RuntimeError, ValueError)):
        find_knn_gpu(torch.zeros(4, 32), torch.zeros(4, 32))
    with pytest.raises(RuntimeError):
        ops.voxelize(np.zeros((10, 3)), 0.05, device='cuda')


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'deepglobalregistration_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert '/root/reference' not in src, f


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """The boundary is a C ABI: include/dgr_hip.h must compile as C99 (no C++ / torch types in the signatures)
    and a C program must link against libdgr_hip.so and call into it (dgr_version needs no GPU)."""
    import os
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, 'deepglobalregistration_amd', 'lib')
    if not os.path.exists(os.path.join(lib_dir, 'libdgr_hip.so')):
        pytest.skip('library not built')
    src = tmp_path / 'c_abi.c'
    src.write_text('#include <stdio.h>\n#include "dgr_hip.h"\n'
                   'int main(void) { dgr_params p; p.max_iter = 1000; (void)p;\n'
                   '  printf("%s\\n", dgr_version()); return dgr_last_error() == 0; }\n')
    exe = tmp_path / 'c_abi'
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-pedantic', '-Werror', '-I', os.path.join(root, 'include'),
                    str(src), '-o', str(exe), '-L', lib_dir, '-ldgr_hip', f'-Wl,-rpath,{lib_dir}',
                    '-Wl,-rpath,/opt/rocm/lib'], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert 'dgr_hip' in out


def test_registration_kernel_has_no_packed_f32_arithmetic(tmp_path):
    """reg.hip is built without the SLP vectoriser (csrc/Makefile: FLAGS_reg): builds of the registration kernel with packed
    f32 arithmetic were not reproducible next to a second process on the GPU (tools/r06_runs/run2.sh,
    run3.sh).  Guard: the object the library links contains the kernel WITHOUT v_pk_{fma,mul,add}_f32."""
    import shutil
    import subprocess
    objdump = '/opt/rocm/lib/llvm/bin/llvm-objdump'
    obj = os.path.join(ROOT, 'deepglobalregistration_amd', 'csrc', 'build', 'reg.o')
    if not (os.path.exists(objdump) and os.path.exists(obj) and shutil.which('hipcc')):
        pytest.skip('no build tree / ROCm tools')
    mk = open(os.path.join(ROOT, 'deepglobalregistration_amd', 'csrc', 'Makefile')).read()
    assert re.search(r'^FLAGS_reg\s*:=.*-fno-slp-vectorize', mk, flags=re.M)
    # the device code object is bundled inside the host object: extract (next to a copy of the object), then disassemble
    shutil.copy(obj, tmp_path / 'reg.o')
    subprocess.run([objdump, '--offloading', 'reg.o'], check=True, capture_output=True, cwd=tmp_path)
    co = [f for f in os.listdir(tmp_path) if 'gfx950' in f]
    assert len(co) == 1, os.listdir(tmp_path)
    asm = subprocess.run([objdump, '-d', co[0]], check=True, capture_output=True, text=True, cwd=tmp_path).stdout
    assert 'registration_kernel' in asm
    packed = re.findall(r'\bv_pk_(?:fma|mul|add)_f32\b', asm)
    assert not packed, f'{len(packed)} packed-f32 instructions in reg.o'
