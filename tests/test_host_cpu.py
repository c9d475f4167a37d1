"""CPU-side checks: C-ABI library exports, header/binding sync, weight ABI, loud failure without a GPU."""
import os
import re

import pytest
import torch

from point_sam_amd import _lib, get_config
from point_sam_amd.weights import check_state_dict, expected_shapes, random_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_has_no_packed_fp32_instruction_with_nonuniform_op_sel():
    """The instruction forms behind round 3's multi-stream corruption (point_sam_amd/isa_lint.py, profiles/r04/r04_hazard.txt) must not ship."""
    from point_sam_amd import isa_lint
    from point_sam_amd.build import build_library
    assert isa_lint.hazardous("v_pk_mul_f32 v[52:53], v[74:75], v[42:43] op_sel:[0,1]")
    assert isa_lint.hazardous("\tv_pk_mul_f32 v[6:7], v[18:19], v[6:7] op_sel:[1,0]     // 000000001234: D3B14006 ...")
    assert isa_lint.hazardous("v_pk_fma_f32 v[52:53], v[74:75], v[42:43], v[60:61] op_sel:[0,1,0] neg_lo:[0,0,1]")
    assert isa_lint.hazardous("v_pk_add_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1] op_sel_hi:[1,0]")
    for safe in ("v_pk_mul_f32 v[2:3], v[4:5], v[6:7]", "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0]", "v_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[1,1]",
                 "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel_hi:[0,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]", "v_pk_mov_b32 v[2:3], v[4:5], v[6:7] op_sel:[1,0]",
                 "v_mul_f32_e32 v52, v74, v43"):
        assert not isa_lint.hazardous(safe), safe
    assert isa_lint.lint(build_library()) == []


def test_library_exports_every_declared_symbol():
    from point_sam_amd.build import build_library
    build_library()
    lib = _lib.load()
    hdr = open(os.path.join(ROOT, "include", "pointsam_hip.h")).read()
    declared = set(re.findall(r"\b(psam_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    # the entry points of the measured-and-rejected paths sit in `#ifdef PSAM_BUILD_EXPERIMENTS` blocks of the header and are exported by experiments
    # builds only (PSAM_BUILD_EXPERIMENTS=1): exactly the names _lib.EXPERIMENTAL lists
    guarded = set(re.findall(r"\b(psam_[a-z0-9_]+)\s*\(", " ".join(re.findall(r"#ifdef PSAM_BUILD_EXPERIMENTS(.*?)#endif", hdr, re.S))))
    assert guarded == set(_lib.EXPERIMENTAL), guarded ^ set(_lib.EXPERIMENTAL)
    for name in declared:
        assert hasattr(lib, name) or (name in guarded and not _lib.has_experiments()), name
    assert lib.psam_version() == 100


def test_invalid_arguments_return_status_not_crash():
    lib = _lib.load()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.psam_fps(None, 1, 16, 4, None, None, None, 0, None) == -1
    assert b"null" in lib.psam_last_error_string()
    assert lib.psam_knn(None, None, 1, 1, 1, 1, None, None) == -1
    assert lib.psam_fps_workspace_bytes(2, 1000, 10) == 2 * 4 * 4096 * 4 + 2 * (2 * 64 * 8 + 16) + 2 * (32768 + 8) * 4   # planar xyz + min-dist + cooperative keys / counter + the pruned kernel's box and cell counters


def test_attention_keysplit_is_per_context():
    """The cap of the single-cloud attention's key split (ops.attention_keysplit: the multi-stream pipelines run with 1, everything else with 4) is a
    context variable like the GEMM mode: another thread -- another pipeline, a server worker -- does not see it."""
    import threading
    from point_sam_amd import ops
    assert ops.current_attention_keysplit() == 4
    with ops.attention_keysplit(1):
        assert ops.current_attention_keysplit() == 1
        seen = []
        t = threading.Thread(target=lambda: seen.append(ops.current_attention_keysplit()))
        t.start(); t.join()
        assert seen == [4]
        with ops.attention_keysplit(0):      # clamped: 1 = never split
            assert ops.current_attention_keysplit() == 1
        assert ops.current_attention_keysplit() == 1
    assert ops.current_attention_keysplit() == 4


def test_product_path_refuses_cpu():
    from point_sam_amd.model import PointCloudSAM
    cfg = get_config("tiny")
    with pytest.raises(RuntimeError):
        PointCloudSAM(cfg, random_state_dict(cfg), device="cpu")
    from point_sam_amd import ops
    with pytest.raises(_lib.PointSamHipError):
        ops.fps(torch.zeros(1, 8, 3), 2)


def test_weight_abi_names():
    for name in ("base", "large", "giant"):
        cfg = get_config(name)
        shapes = expected_shapes(cfg)
        assert "pc_encoder.patch_embed.patch_encoder.conv2.3.weight" in shapes
        assert "mask_decoder.transformer.final_attn_token_to_image.out_proj.weight" in shapes
        assert shapes["pc_encoder.out_proj.weight"] == (256, cfg.vit.dim)
    cfg = get_config("large")
    assert expected_shapes(cfg)["pc_encoder.transformer.blocks.23.mlp.fc1_g.weight"] == (2730, 1024)
    n = sum(int(torch.tensor(s).prod()) for s in expected_shapes(cfg).values())
    assert 300e6 < n < 330e6  # ViT-L + decoder
    cfg = get_config("tiny")
    sd = random_state_dict(cfg)
    sd["pc_encoder.transformer.cls_token"] = torch.zeros(1, 1, 64)  # timm leftovers are tolerated
    check_state_dict(cfg, sd)
    sd["bogus.weight"] = torch.zeros(1)
    with pytest.raises(KeyError):
        check_state_dict(cfg, sd)


def test_checkpoint_round_trip_through_safetensors(tmp_path):
    """evaluation/inference.py:46 loads `model.safetensors` by name: a file with the reference's names (+ the timm leftovers a real
    checkpoint carries, in fp16 like some published ones) loads into the name->fp32 dict the model consumes; a missing or mis-shaped
    tensor is an error."""
    from safetensors.torch import save_file
    from point_sam_amd.weights import load_safetensors
    cfg = get_config("tiny")
    sd = random_state_dict(cfg, seed=5)
    extra = {"pc_encoder.transformer.cls_token": torch.zeros(1, 1, 64), "pc_encoder.transformer.pos_embed": torch.zeros(1, 5, 64)}
    path = str(tmp_path / "model.safetensors")
    save_file({k: v.half().contiguous() for k, v in {**sd, **extra}.items()}, path)
    got = load_safetensors(cfg, path)
    assert all(got[k].dtype == torch.float32 and torch.equal(got[k], sd[k].half().float()) for k in sd)
    bad = dict(sd); bad.pop("pc_encoder.out_proj.weight")
    save_file({k: v.contiguous() for k, v in bad.items()}, path)
    with pytest.raises(KeyError):
        load_safetensors(cfg, path)
    bad = dict(sd); bad["pc_encoder.out_proj.weight"] = torch.zeros(3, 3)
    save_file({k: v.contiguous() for k, v in bad.items()}, path)
    with pytest.raises((ValueError, KeyError)):
        load_safetensors(cfg, path)


def test_synthetic_inputs_match_the_oracles_generator():
    """bench.py draws its inputs from the package (it may touch oracle/ only for the cpu_baseline leg): same generator as the oracle's."""
    from oracle import pointsam_oracle as O
    from point_sam_amd.synthetic import synthetic_batch
    for a, b in zip(synthetic_batch(2, 500, seed=7, num_prompts=2), O.synthetic_batch(2, 500, seed=7, num_prompts=2)):
        assert torch.equal(a, b)
    xyz = synthetic_batch(3, 1000)[0]
    assert torch.allclose(xyz.norm(dim=2).max(dim=1).values, torch.ones(3))


def test_mlp3_weight_stack_and_ln_bound():
    """Host-side preparation for psam_mlp3 (stacked [M, out, in] weights, shape chaining) and the LayerNorm output bound that scales the
    packed output of a row-LayerNorm epilogue (|LN(x) gamma + beta| <= sqrt(n - 1) max|gamma| + max|beta|)."""
    import torch
    from point_sam_amd import ops
    g = torch.Generator().manual_seed(0)
    mk = lambda i, o: (torch.randn(o, i, generator=g), torch.randn(o, generator=g))
    mw = ops.Mlp3Weights([[mk(8, 12), mk(12, 12), mk(12, 5)] for _ in range(3)])
    assert (mw.M, mw.din, mw.dh, mw.dout) == (3, 8, 12, 5) and mw.w1.shape == (3, 12, 8) and mw.w3.shape == (3, 5, 12) and mw.b3.shape == (3, 5)
    with pytest.raises(ValueError):
        ops.Mlp3Weights([[mk(8, 12), mk(10, 12), mk(12, 5)]])
    gam, bet = torch.randn(256, generator=g), torch.randn(256, generator=g)
    x = torch.randn(1000, 256, generator=g) * torch.logspace(-3, 3, 1000)[:, None]
    x[0] = 0; x[0, 7] = 1e6      # one-hot row: the extreme case of the bound
    y = torch.nn.functional.layer_norm(x.double(), (256,), gam.double(), bet.double(), 1e-5)
    assert y.abs().max().item() <= ops.row_ln_bound(gam, bet) * (1 + 1e-6)
    assert y[0].abs().max().item() > 0.9 * (255 ** 0.5) * gam[7].abs().item() - bet.abs().max().item()


def test_ctypes_structs_match_the_header(tmp_path):
    """Every struct of include/pointsam_hip.h that the Python host mirrors with ctypes has the same size (and, for the ones with mixed field
    types, the same offsets of a few late fields) as the C compiler gives it: an edit to one side without the other fails here, not on the GPU."""
    import ctypes
    import shutil
    import subprocess
    from point_sam_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("needs gcc")
    pairs = [("psam_gemm_fuse_t", _lib.GemmFuse, "out_bound"), ("psam_twoway_tokens_t", _lib.TwoWayTokens, "ws_floats"),
             ("psam_eva_block_weights_t", _lib.EvaBlockWeights, "eps"), ("psam_eva_block_plan_t", _lib.EvaBlockPlan, "o_lnd"),
             ("psam_eva_gelu_block_weights_t", _lib.EvaGeluBlockWeights, "eps"), ("psam_eva_gelu_block_plan_t", _lib.EvaGeluBlockPlan, "o_s2"),
             ("psam_skinny_job_t", _lib.SkinnyJob, "act"), ("psam_skinny_jobs_t", _lib.SkinnyJobs, "n"), ("psam_mlp3_args_t", _lib.Mlp3Args, "dout"),
             ("psam_patch_encoder_weights_t", _lib.PatchEncoderWeights, "eps"), ("psam_patch_encoder_plan_t", _lib.PatchEncoderPlan, "o_s23"),
             ("psam_upscale_weights_t", _lib.UpscaleWeights, "eps"), ("psam_upscale_plan_t", _lib.UpscalePlan, "o_s3"),
             ("psam_attn_weights_t", _lib.AttnWeights, "o_b"), ("psam_twoway_layer_weights_t", _lib.TwoWayLayerW, "m2_b"),
             ("psam_twoway_weights_t", _lib.TwoWayWeights, "nf_b"), ("psam_twoway_plan_t", _lib.TwoWayPlan, "o_cat_bias")]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "pointsam_hip.h"', 'int main(void) {']
    for name, _, field in pairs:
        src.append(f'    printf("{name} %zu %zu\\n", sizeof({name}), offsetof({name}, {field}));')
    src += ['    return 0;', '}']
    c = tmp_path / "sizes.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = dict((l.split()[0], (int(l.split()[1]), int(l.split()[2]))) for l in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.splitlines())
    for name, cls, field in pairs:
        assert out[name] == (ctypes.sizeof(cls), getattr(cls, field).offset), (name, out[name], ctypes.sizeof(cls), getattr(cls, field).offset)


def test_host_side_planning_functions_without_a_gpu():
    """Host logic of the library that needs no device: the split-K factor per shape, workspace / prepared-blob sizes of the coarse entry points."""
    from point_sam_amd import _lib
    lib = _lib.load()
    sk = lib.psam_gemm_f16x3p_splitk
    assert sk(512, 1408, 6144, 0) == 4 and sk(512, 1408, 1408, 1) in (3, 4)      # fc2 / proj of the giant encoder at one cloud: few tiles, long K
    assert sk(4096, 1024, 1024, 0) == 1                                          # a batch of clouds: never (M > 2048)
    assert sk(512, 6144, 1408, 0) == 1 and sk(512, 1408, 6144, 3) == 1 and sk(512, 1408, 512, 0) == 1      # enough tiles / SwiGLU / short K
    for M in (256, 4096):
        a, b = lib.psam_eva_block_ws_bytes(M, 1024, 2730), lib.psam_eva_block_ws_bytes(2 * M, 1024, 2730)
        assert 0 < a < b and a % 256 == 0
    assert lib.psam_eva_block_prepared_bytes(1024, 2730) > 4 * (3 * 1024 * 1024 + 1024 * 1024 + 2 * 2752 * 1024 + 1024 * 2752)
    assert lib.psam_eva_block_ws_bytes(0, 1024, 2730) == 0 and lib.psam_eva_block_prepared_bytes(0, 1) == 0
    assert lib.psam_patch_encoder_ws_bytes(8 * 512 * 64, 8 * 512, 128, 512) > 4 * 8 * 512 * 64 * (128 + 128 + 512)
    assert lib.psam_twoway_decoder_prepared_bytes(2, 256, 2048, 2) > 4 * 2 * (4 * 256 * 256 + 8 * 128 * 256 + 2 * 2048 * 256)
    assert lib.psam_twoway_decoder_ws_bytes(8, 6, 512, 256, 2048) > 4 * 6 * 8 * 512 * 256
    assert lib.psam_upscale_masks_ws_bytes(8, 32768, 512, 3, 256) > 4 * 8 * 32768 * 256
    if _lib.has_experiments():
        assert lib.psam_twoway_tokens_ws_floats(2048) == 64 * (5 * 256 + 2048) + 64


def test_gemm_mode_is_per_context():
    """The GEMM arithmetic mode is a context variable: a model (or server thread) entering its own precision does not change what another
    thread sees (two models of different precision, the threaded demo server)."""
    import threading
    from point_sam_amd import ops
    seen, go, done = {}, threading.Event(), threading.Event()

    def other():
        with ops.gemm_mode("bf16x6"):
            seen["inside_other"] = ops.GEMM_MODE
            go.set()
            done.wait(5)
            seen["other_after_main_changed"] = ops.current_gemm_mode()

    t = threading.Thread(target=other)
    t.start()
    go.wait(5)
    assert ops.GEMM_MODE == "f32"                      # untouched by the other thread
    with ops.gemm_mode("f16x3"):
        assert ops.current_gemm_mode() == "f16x3"
        with ops.gemm_mode("f32"):
            assert ops.GEMM_MODE == "f32"
        assert ops.GEMM_MODE == "f16x3"
        done.set()
        t.join()
    assert seen == {"inside_other": "bf16x6", "other_after_main_changed": "bf16x6"} and ops.GEMM_MODE == "f32"
    import pytest
    with pytest.raises(ValueError):
        ops.gemm_mode("fp8")


def test_library_keeps_no_stream_state_and_allocates_only_in_prepare():
    """VERDICT r05 item 7 / the boundary's own contract ("no allocation inside ... thread-safe given distinct streams/workspaces"): the product
    sources create no streams or events, allocate device memory only in a *_prepare entry (a staging buffer it frees), and keep no (device, stream) tables;
    the arrival counters of the in-kernel fix-ups are the caller's (PSAM_COUNTER_BYTES), and the entry points that used to manage them are gone."""
    from point_sam_amd import ops
    csrc = os.path.join(ROOT, "point_sam_amd", "csrc")
    hdr = open(os.path.join(ROOT, "include", "pointsam_hip.h")).read()
    assert int(re.search(r"#define PSAM_COUNTER_BYTES (\d+)", hdr).group(1)) == ops.COUNTER_BYTES
    for gone in ("psam_stream_has_arrival_counters", "psam_gemm_f16x3p_reset_splitk_state", "psam_attention_f16x3_reset_keysplit_state"):
        assert gone not in hdr and gone not in _lib.SIGNATURES
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".cpp")):
            continue
        src = open(os.path.join(csrc, f)).read()
        default = re.sub(r"#ifdef PSAM_BUILD_EXPERIMENTS.*?#endif", "", src, flags=re.S)      # what a default build compiles
        default = re.sub(r"#ifndef PSAM_BUILD_EXPERIMENTS(.*?)#else.*?#endif", r"\1", default, flags=re.S)
        assert "hipStreamCreate" not in default and "hipEventCreate" not in default, f
        assert "std::map" not in default and "std::mutex" not in default, f
        for m in re.finditer(r"hipMalloc\w*\(", default):
            fn = re.findall(r"PSAM_API \w+ (psam_\w+)\(", default[:m.start()])[-1]
            assert fn.endswith("_prepare"), (f, fn)
    # the host's blocks: one per use_counters scope, validated
    with pytest.raises(ValueError):
        ops.use_counters(torch.zeros(4, dtype=torch.int32))
    from point_sam_amd.streams import pipeline_streams_report
    assert isinstance(pipeline_streams_report(), dict)
