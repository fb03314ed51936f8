"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/conmamba_hip.h declares; argument validation works without a GPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def native():
    import mamba_asr_amd._native as N
    if not os.path.exists(N.LIB_PATH):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "mamba_asr_amd", "csrc")])
    return N


def test_header_symbols_exported(native):
    hdr = open(os.path.join(ROOT, "include", "conmamba_hip.h")).read()
    hdr = re.sub(r"#ifdef CM_ABLATE.*?#endif", "", hdr, flags=re.S)      # the ablation build's switch is not part of the product ABI
    declared = set(re.findall(r"^\s*(?:int|int32_t|int64_t|const char \*)\s*\*?\s*(cm_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert {"cm_selective_scan_fwd", "cm_selective_scan_bwd", "cm_causal_conv1d_fwd", "cm_causal_conv1d_bwd",
            "cm_abi_version", "cm_last_error", "cm_scan_num_chunks", "cm_scan_cl_fwd", "cm_conv_cl_fwd", "cm_conv_xproj", "cm_fbank_wav", "cm_cnn_front", "cm_ln_pw_glu", "cm_dwconv1d_fwd", "cm_dwconv1d_bwd", "cm_dwconv_cl_fwd", "cm_dwconv_cl_bwd", "cm_dwconv_cl_workspace_floats", "cm_causal_conv1d_update", "cm_selective_state_update", "cm_add_layernorm", "cm_layernorm_fwd", "cm_layernorm_bwd", "cm_layernorm_bwd_workspace_floats",
            "cm_glu_dwconv_ln_gelu", "cm_cnn_block1", "cm_gemm_bf16", "cm_fbank_mel_db", "cm_fbank_finish",
            "cm_spec_drop"} <= declared
    handle = C.CDLL(native.LIB_PATH)
    for name in declared:
        assert hasattr(handle, name), f"{name} declared in conmamba_hip.h but not exported"
    assert {s[0] for s in native.SYMBOLS} == declared
    # the product library carries no process-global switch (ablation variants live in the CM_ABLATE build only)
    for name in ("cm_debug_set", "cm_debug_get", "cm_scan_set_split"):
        assert not hasattr(handle, name), f"{name} must not be exported by the product library"


def test_abi_version_and_chunks(native):
    lib = native.lib()
    assert lib.cm_abi_version() == native.ABI_VERSION
    assert lib.cm_scan_num_chunks(1) == 1
    assert lib.cm_scan_num_chunks(64) == 1
    assert lib.cm_scan_num_chunks(65) == 2


def test_scan_chunk_policy_and_workspace_are_host_functions(native):
    """cm_scan_cl_fwd_auto_chunks / cm_scan_cl_fwd_workspace_bytes depend on sizes only (no GPU): large launches are not
    cut, launches of fewer than 256 workgroups are cut to ~1024 with chunks of at least 128 steps, and the workspace is
    three (ndir, batch, chunks, dim, 16) fp32 slabs for the chunk count the launch will really use."""
    lib = native.lib()
    auto = lib.cm_scan_cl_fwd_auto_chunks
    assert auto(64, 1000, 512, 2) == 1 and auto(16, 1000, 512, 2) == 1          # >= 256 workgroups
    assert auto(8, 1000, 512, 2) == 7                                           # 1024 // 128 = 8, capped at 1000 // 128
    assert auto(4, 4000, 1024, 2) == 8 and auto(1, 4000, 512, 2) == 31
    assert auto(1, 100, 64, 1) == 1 and auto(0, 100, 64, 1) == 1
    a = native.ScanClArgs()
    a.batch, a.seqlen, a.dim, a.ndir = 4, 4000, 1024, 2
    assert lib.cm_scan_cl_fwd_workspace_bytes(C.byref(a)) == 0                  # time_chunks 0 / 1: none
    a.time_chunks = 8                                                           # 4000 / 8 = 500 -> 512-step chunks -> 8 chunks
    assert lib.cm_scan_cl_fwd_workspace_bytes(C.byref(a)) == 3 * 2 * 4 * 8 * 1024 * 16 * 4
    a.seqlen, a.time_chunks = 50, 7                                             # 16-step chunks -> 4 chunks, not 7
    assert lib.cm_scan_cl_fwd_workspace_bytes(C.byref(a)) == 3 * 2 * 4 * 4 * 1024 * 16 * 4
    assert lib.cm_scan_cl_fwd_workspace_bytes(None) == 0
    # the backward's policy: under 512 workgroups (two per CU) it cuts to about 1024, chunks of at least 128 steps
    bauto = lib.cm_scan_cl_bwd_auto_chunks
    assert bauto(32, 1000, 512, 2) == 1 and bauto(64, 1000, 512, 2) == 1
    assert bauto(4, 4000, 1024, 2) == 8 and bauto(16, 1000, 512, 2) == 4 and bauto(1, 100, 64, 1) == 1
    b = native.ScanClBwdArgs()
    b.batch, b.seqlen, b.dim, b.ndir, b.time_chunks = 4, 4000, 1024, 2, 1
    b.dir[0].dt_rank = b.dir[1].dt_rank = 32
    one = lib.cm_scan_cl_bwd_workspace_bytes(C.byref(b))
    b.time_chunks = 0
    assert lib.cm_scan_cl_bwd_workspace_bytes(C.byref(b)) - one == 4 * 2 * 4 * 1024 * 7 * (16 + 32 + 4 + 3 * 16)


def test_bad_arguments_are_rejected_not_fatal(native):
    lib = native.lib()
    a = native.ScanFwdArgs()          # all zero: sizes invalid
    rc = lib.cm_selective_scan_fwd(C.byref(a))
    assert rc == -1
    assert b"bad sizes" in lib.cm_last_error()
    a.batch, a.dim, a.seqlen, a.dstate = 1, 8, 16, 16
    rc = lib.cm_selective_scan_fwd(C.byref(a))   # null pointers
    assert rc == -1 and b"non-NULL" in lib.cm_last_error()
    assert lib.cm_selective_scan_fwd(None) == -1
    c = native.ConvArgs()
    assert lib.cm_causal_conv1d_fwd(C.byref(c)) == -1
    b = native.ScanBwdArgs()
    assert lib.cm_selective_scan_bwd(C.byref(b)) == -1


def test_ops_refuse_cpu_tensors(native):
    import torch
    from mamba_asr_amd import ops
    x = torch.zeros(1, 8, 16)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.causal_conv1d_fwd(x, torch.zeros(8, 4))


def test_cast_cache_follows_in_place_updates(native):
    """ops.cast_cached (the bf16 weight copies the autograd nodes reuse within a step) must notice every in-place
    update of the parameter: optimizer steps, copy_ (load_state_dict) and updates through detach()."""
    import torch
    from mamba_asr_amd import ops
    p = torch.nn.Parameter(torch.randn(8, 8))
    c1 = ops.cast_cached(p, torch.bfloat16)
    assert ops.cast_cached(p, torch.bfloat16) is c1 and not c1.requires_grad
    opt = torch.optim.SGD([p], lr=0.5)
    p.grad = torch.ones_like(p)
    opt.step()
    c2 = ops.cast_cached(p, torch.bfloat16)
    assert c2 is not c1 and torch.equal(c2, p.detach().bfloat16())
    with torch.no_grad():
        p.copy_(torch.zeros(8, 8))
    assert torch.equal(ops.cast_cached(p, torch.bfloat16), torch.zeros(8, 8, dtype=torch.bfloat16))
    p.detach().add_(1.0)
    assert torch.equal(ops.cast_cached(p, torch.bfloat16), torch.ones(8, 8, dtype=torch.bfloat16))
    assert ops.cast_cached(p, torch.float32).data_ptr() == p.data_ptr()
