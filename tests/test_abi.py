"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/mtn_hip.h declares,
ctypes mirrors have the C struct sizes, and product code never reaches into oracle/."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib_or_build():
    from mtn_amd import build, lib
    if not os.path.exists(lib.LIB_PATH):
        build.build(verbose=False)
    return lib


def test_library_exports_every_declared_symbol():
    lib = _lib_or_build()
    hdr = open(os.path.join(ROOT, "include", "mtn_hip.h")).read()
    declared = set(re.findall(r"\b(mtn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(lib.SYMBOLS), (declared ^ set(lib.SYMBOLS))
    h = lib.load()
    for name in declared:
        assert getattr(h, name) is not None
    assert h.mtn_version() >= 113


def test_struct_layouts_match_c(tmp_path):
    lib = _lib_or_build()
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "mtn_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(mtn_dropout),'
                   ' sizeof(mtn_gemm_problem), sizeof(mtn_attn_args), sizeof(mtn_mha_args), sizeof(mtn_ffn_args), sizeof(mtn_ln_epilogue),'
                   ' sizeof(mtn_ln_fold_desc), sizeof(mtn_transpose_desc), sizeof(mtn_tt_ln_unit), sizeof(mtn_tt_aux), sizeof(mtn_decode_stage), sizeof(mtn_decode_args), sizeof(mtn_beam_args));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    mine = [ctypes.sizeof(c) for c in (lib.Dropout, lib.GemmProblem, lib.AttnArgs, lib.MhaArgs, lib.FfnArgs, lib.LnEpilogue, lib.LnFoldDesc, lib.TransposeDesc,
                                       lib.TtLnUnit, lib.TtAux, lib.DecodeStage, lib.DecodeArgs, lib.BeamArgs)]
    assert sizes == mine


def test_bad_arguments_are_reported_not_crashed():
    lib = _lib_or_build()
    h = lib.load()
    rc = h.mtn_layernorm_fwd(0, 0, 512, 1e-6, None, None, None, None, None, None, None, None)
    assert rc != 0 and b"mtn_layernorm_fwd" in h.mtn_last_error()
    rc = h.mtn_gemm(7, 1, None, None)
    assert rc != 0


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "mtn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "/root/reference" not in text, f


def test_ops_refuse_cpu_tensors():
    import torch
    from mtn_amd import ops
    with pytest.raises(Exception):
        ops.layer_norm(torch.randn(4, 64), torch.ones(64), torch.zeros(64))


def test_bench_line_is_terse_enough_for_the_drivers_record():
    """bench.py prints the terse form of its record: the driver keeps an 8 KB tail of stdout, so the WHOLE line must fit in it (< 7 KB) and still
    carry the contract's keys plus every secondary figure README.md quotes.  Checked on a committed verbose record of a real run
    (profiles/r06_bench_full.json, written by `bench.py --full-record`)."""
    import json
    import sys
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")))
    line = bench.terse_line(full)
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 7000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(line["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
    assert "workload" in line["config"] and "model" not in line["config"]
    sec = line["secondary"]
    assert {"batch64_one_gpu", "dp_schedule_one_rank", "batch_sweep", "cfg4_long_context", "corpus_loop", "decode"} <= set(sec)
    assert sec["batch64_one_gpu"]["samples_per_s"] > 0 and sec["decode"]["beam"]["hypothesis_tokens_per_s"] > 0
    assert list(line)[-1] == "secondary", "the secondary figures go last: the driver's record keeps the TAIL of the line"
