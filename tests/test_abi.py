"""CPU: the C-ABI library loads and exports every symbol include/srt.h declares; without a GPU every
compute entry point fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "srt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(srt_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for s in ("srt_trace_batch", "srt_trace_batch_device", "srt_plasma_params", "srt_model_create_ngo",
              "srt_model_create_interp", "srt_model_create_interp_file", "srt_model_create_scattered_file", "srt_model_create_scattered_file_root",
              "srt_write_ray_file", "srt_read_rays_file"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from stanford_raytracer_amd import api

    lib = api.lib()
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof of srt_params as the C compiler lays out include/srt.h == the ctypes mirror."""
    import subprocess

    from stanford_raytracer_amd import api

    names = [f[0] for f in api.Params._fields_]
    cnames = [("del" if n == "del_" else n) for n in names]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "srt.h"\nint main(void){printf("%zu", sizeof(srt_params));'
                   + "".join('printf(" %%zu", offsetof(srt_params, %s));' % n for n in cnames) + "return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == C.sizeof(api.Params)
    assert got[1:] == [getattr(api.Params, n).offset for n in names]
    p = api.make_params(maxsteps=256, outputper=16)
    assert api.lib().srt_rows_per_ray(C.byref(p)) == 16
    p = api.make_params(maxsteps=10, outputper=3)
    assert api.lib().srt_rows_per_ray(C.byref(p)) == 4


def test_no_cpu_fallback():
    import torch

    from stanford_raytracer_amd import api

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.SrtError, match="no HIP device"):
        api.init(0)
    with pytest.raises(api.SrtError):
        api.Model.ngo("/nonexistent/newray.in")


def test_product_does_not_touch_the_oracle():
    pkg = os.path.join(ROOT, "stanford_raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h", ".f90")):
                text = open(os.path.join(dirpath, f)).read()
                assert "srt_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
