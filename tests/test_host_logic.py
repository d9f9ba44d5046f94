"""Host-side behaviour that needs no kernels: config, module tree / state-dict keys, weight tying, init flags,
HF Auto registration, error conventions of the reference."""
import json

import pytest
import torch

from caduceus_amd import (BiMambaWrapper, CaduceusConfig, CaduceusForMaskedLM, CaduceusForSequenceClassification,
                          register_auto_classes)
from caduceus_amd.mamba import Mamba
from conftest import MODEL_VARIANTS, load_golden_model


@pytest.mark.parametrize("name", MODEL_VARIANTS)
def test_state_dict_keys_and_shapes_match_reference(name):
    cfg, sd, _ = load_golden_model(name)
    model = CaduceusForMaskedLM(CaduceusConfig(**cfg))
    ours = model.state_dict()
    # same key set incl. the duplicate tied aliases (order inside a Mamba follows upstream mamba_ssm)
    assert sorted(ours.keys()) == sorted(sd.keys())
    for k in sd:
        assert tuple(ours[k].shape) == tuple(sd[k].shape), k
        assert ours[k].dtype == sd[k].dtype, k


def test_weight_tying_and_flags():
    cfg, _, _ = load_golden_model("ps_fused")
    model = CaduceusForMaskedLM(CaduceusConfig(**cfg))
    emb = model.get_input_embeddings()
    assert model.lm_head.weight is emb.weight  # modeling_caduceus.py:434-439
    mix = model.caduceus.backbone.layers[0].mixer.submodule
    assert mix.mamba_rev.in_proj.weight is mix.mamba_fwd.in_proj.weight  # :114-118
    assert mix.mamba_rev.out_proj.weight is mix.mamba_fwd.out_proj.weight
    assert mix.mamba_rev.x_proj.weight is not mix.mamba_fwd.x_proj.weight
    assert getattr(mix.mamba_fwd.A_log, "_no_weight_decay") and getattr(mix.mamba_fwd.D, "_no_weight_decay")
    assert getattr(mix.mamba_fwd.dt_proj.bias, "_no_reinit")
    # dt_proj.bias survives _init_weights (the reference zeroes every other Linear bias, :318-321)
    assert float(mix.mamba_fwd.dt_proj.bias.abs().min()) > 0
    n_params = sum(p.numel() for p in model.parameters())
    assert n_params == sum(p.numel() for p in set(model.parameters()))


def test_param_count_of_released_model_shape():
    """7 725 312 parameters at d_model 256 / n_layer 16 for both Ph and PS."""
    ssm = dict(d_state=16, d_conv=4, expand=2, dt_rank="auto")
    comp = {i: i for i in range(12)}
    for rcps in (True, False):
        with torch.device("meta"):
            m = CaduceusForMaskedLM(CaduceusConfig(d_model=256, n_layer=16, vocab_size=12, ssm_cfg=ssm, rcps=rcps,
                                                   complement_map=dict(comp)))
        assert sum(p.numel() for p in m.parameters()) == 7_725_312


def test_vocab_and_complement_padding():
    cfg, _, _ = load_golden_model("ps_fused")
    c = CaduceusConfig(**cfg)
    assert c.vocab_size == 12
    m = CaduceusForMaskedLM(c)
    assert m.config.vocab_size == 16 and len(m.config.complement_map) == 16  # modeling_caduceus.py:352-357
    assert m.config.complement_map[13] == 13 and m.config.complement_map[7] == 10


def test_config_roundtrip_json():
    cfg, _, _ = load_golden_model("ps_fused")
    c = CaduceusConfig(**cfg)
    c2 = CaduceusConfig(**json.loads(c.to_json_string()))
    assert c2.complement_map == c.complement_map and c2.model_type == "caduceus"
    assert all(isinstance(k, int) for k in c2.complement_map)


def test_reference_error_conventions():
    with pytest.raises(NotImplementedError):
        BiMambaWrapper(32, bidirectional_strategy="concat")  # modeling_caduceus.py:101-102
    cfg, _, _ = load_golden_model("ps_fused")
    with pytest.raises(AssertionError):
        CaduceusForMaskedLM(CaduceusConfig(**{**cfg, "complement_map": None}))  # :350
    m = CaduceusForMaskedLM(CaduceusConfig(**cfg))
    with pytest.raises(NotImplementedError):
        m.set_input_embeddings(torch.nn.Embedding(16, 32))  # :421-422
    with pytest.raises(NotImplementedError):
        m.set_output_embeddings(torch.nn.Linear(32, 16))
    with pytest.raises(NotImplementedError):
        CaduceusForSequenceClassification(CaduceusConfig(**cfg), pooling_strategy="median")


def test_mamba_constructor_signature_and_init():
    torch.manual_seed(0)
    m = Mamba(64, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=1e-3, dt_max=0.1, dt_init="random",
              dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True, layer_idx=3)
    assert m.dt_rank == 4 and m.d_inner == 128
    assert torch.allclose(m.A_log.exp(), torch.arange(1, 17).float().repeat(128, 1))
    dt = torch.nn.functional.softplus(m.dt_proj.bias)
    assert float(dt.min()) >= 1e-4 - 1e-9 and float(dt.max()) <= 0.1 + 1e-6
    assert float(m.dt_proj.weight.abs().max()) <= 4 ** -0.5 + 1e-6


def test_auto_registration_offline(tmp_path):
    from transformers import AutoConfig, AutoModelForMaskedLM
    register_auto_classes()
    register_auto_classes()  # idempotent
    cfg, sd, _ = load_golden_model("ps_fused")
    model = CaduceusForMaskedLM(CaduceusConfig(**cfg))
    model.load_state_dict(sd)
    model.save_pretrained(tmp_path)
    c = AutoConfig.from_pretrained(tmp_path)
    assert isinstance(c, CaduceusConfig)
    m2 = AutoModelForMaskedLM.from_pretrained(tmp_path)
    assert isinstance(m2, CaduceusForMaskedLM)
    for k, v in model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k


def test_forward_signature_has_no_state_param():
    """LMTask.forward inspects the model's forward signature (src/tasks/tasks.py:183-192)."""
    import inspect
    params = inspect.signature(CaduceusForMaskedLM.forward).parameters
    assert "state" not in params
    assert list(params)[1:] == ["input_ids", "inputs_embeds", "labels", "loss_weights", "output_hidden_states",
                                "return_dict"]


def test_tokenizer_ids_and_complement_map():
    """SURVEY.md Appendix A (tokenization_caduceus.py:49-66)."""
    from caduceus_amd import CaduceusTokenizer
    tok = CaduceusTokenizer(model_max_length=64)
    assert tok.vocab_size == 12
    assert tok.get_vocab() == {"[CLS]": 0, "[SEP]": 1, "[BOS]": 2, "[MASK]": 3, "[PAD]": 4, "[RESERVED]": 5, "[UNK]": 6,
                               "A": 7, "C": 8, "G": 9, "T": 10, "N": 11}
    assert tok.complement_map == {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 10, 8: 9, 9: 8, 10: 7, 11: 11}
    ids = tok("acgtnX", add_special_tokens=False)["input_ids"]
    assert ids == [7, 8, 9, 10, 11, 6]
    assert tok.padding_side == "left" and tok.pad_token_id == 4 and tok.mask_token_id == 3
    assert tok.build_inputs_with_special_tokens([7, 8]) == [7, 8, 1]
    CaduceusTokenizer(model_max_length=16, add_special_tokens=False)  # the reference's constructor call (genomics.py:108-111)


def test_chunked_weight_gradient_gemms_match_plain_mm(backend):
    """caduceus_amd/mixer.py: the K-chunked strided-batch formulation of the weight gradients is the same product (fp32 operands: the
    batched cad_gemm_f32 since round 6 -- on the emulator here, on the device under -m gpu; bf16 operands: the library bmm)."""
    from caduceus_amd import mixer
    name, dev = backend
    g = torch.Generator().manual_seed(0)
    T = 8192
    assert mixer._kchunks(T) == 8 and mixer._kchunks(262144) == 64 and mixer._kchunks(100) == 1
    a, b_cm, b_tm = torch.randn(24, T, generator=g), torch.randn(10, T, generator=g), torch.randn(T, 12, generator=g)
    torch.testing.assert_close(mixer._wgrad_cm_cm(a.to(dev), b_cm.to(dev)).cpu(), a @ b_cm.t(), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(mixer._wgrad_cm_tm(a.to(dev), b_tm.to(dev)).cpu(), a @ b_tm, rtol=1e-4, atol=1e-3)
    ab, bb = a.to(torch.bfloat16), b_cm.to(torch.bfloat16)
    torch.testing.assert_close(mixer._wgrad_cm_cm(ab.to(dev), bb.to(dev)).cpu(), ab.float() @ bb.float().t(), rtol=2e-2, atol=0.5)


def test_bench_quotes_a_counter_profile_only_for_the_profiled_scan_sources():
    """bench.pmc_quotable: roofline.traffic comes from profiles/r04_scan_pmc.json only when that profile was taken on the loaded build, or
    on a build with the same scan sources while the loaded library is the build of this tree."""
    import importlib.util
    import os
    from caduceus_amd import _build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cur = "caduceus_amd 0.1.0 (hip gfx950) src " + _build.source_hash()
    pmc = {"lib_version": "caduceus_amd 0.1.0 (hip gfx950) src 000000000000", "scan_src": _build.scan_source_hash()}
    assert bench.pmc_quotable(dict(pmc, lib_version=cur), cur) == "this build (" + cur + ")"
    assert "same scan sources" in bench.pmc_quotable(pmc, cur)
    assert bench.pmc_quotable(dict(pmc, scan_src="ffffffffffff"), cur) is None                    # other scan sources
    assert bench.pmc_quotable(pmc, "caduceus_amd 0.1.0 (hip gfx950) src 111111111111") is None  # a stale library
    assert bench.pmc_quotable({"lib_version": pmc["lib_version"]}, cur) is None                   # an unstamped profile
    # the round's committed profile is quotable for the committed sources (a profile of earlier scan sources is simply not quoted:
    # bench.py then reports traffic = null with the reason)
    path = os.path.join(ROOT, bench.SCAN_PMC_FILE)
    if os.path.exists(path):
        assert bench.pmc_quotable(json.load(open(path)), cur), bench.SCAN_PMC_FILE + " does not belong to the scan sources in the tree"
    old = json.load(open(os.path.join(ROOT, "profiles", "r04_scan_pmc.json")))
    assert old["scan_src"] != _build.scan_source_hash() and bench.pmc_quotable(old, cur) is None  # (round 5 changed both scan kernels)


# ---- the reference's import name (north_star: "train.py and the HF AutoModel path load it unchanged") ---------------------
REGISTRY_MODEL_STRING = "caduceus.modeling_caduceus.CaduceusForMaskedLM"       # /root/reference/src/utils/registry.py:29
HYDRA_CONFIG_TARGET = "caduceus.configuration_caduceus.CaduceusConfig"        # /root/reference/configs/model/caduceus.yaml:4


def _locate(dotted):
    """What the reference does with such a string (src/utils/registry.py + src/utils/config.py `instantiate`: hydra.utils.get_class ==
    import the module part, getattr the last component)."""
    import importlib
    mod, name = dotted.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


def test_reference_import_name_resolves_to_this_engine():
    import os
    ref = "/root/reference"
    if os.path.isdir(ref):  # build container: take the literal strings from the reference's own files (absent on the GPU box)
        reg = open(os.path.join(ref, "src", "utils", "registry.py")).read()
        yml = open(os.path.join(ref, "configs", "model", "caduceus.yaml")).read()
        assert f'"caduceus_lm": "{REGISTRY_MODEL_STRING}"' in reg
        assert f"_target_: {HYDRA_CONFIG_TARGET}" in yml
    import caduceus
    import caduceus_amd
    assert _locate(REGISTRY_MODEL_STRING) is caduceus_amd.CaduceusForMaskedLM
    assert _locate(HYDRA_CONFIG_TARGET) is caduceus_amd.CaduceusConfig
    # not copies: the same module objects under both names, the reference's package exports (caduceus/__init__.py:5-7) included
    for sub in ("configuration_caduceus", "modeling_caduceus", "modeling_rcps", "tokenization_caduceus"):
        assert getattr(caduceus, sub) is getattr(caduceus_amd, sub)
        import importlib
        assert importlib.import_module(f"caduceus.{sub}") is getattr(caduceus_amd, sub)
    for name in ("CaduceusConfig", "Caduceus", "CaduceusForMaskedLM", "CaduceusForSequenceClassification", "CaduceusTokenizer"):
        assert getattr(caduceus, name) is getattr(caduceus_amd, name)
    # the instantiation train.py performs: registry class + yaml config -> a model with the reference's state-dict keys
    cfg, sd, _ = load_golden_model("ps_fused")
    model = _locate(REGISTRY_MODEL_STRING)(_locate(HYDRA_CONFIG_TARGET)(**cfg))
    assert sorted(model.state_dict().keys()) == sorted(sd.keys())


def test_tool_scripts_reference_existing_scripts():
    """The evidence scripts under tools/ call each other (gpu_final.sh -> prof_scan.sh -> ...): a pruned helper must not leave a caller
    behind (tools/prof_step_pmc.sh lost tools/step_families.py that way for a while in round 4)."""
    import glob
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    missing = []
    for f in glob.glob(os.path.join(root, "tools", "*.sh")) + glob.glob(os.path.join(root, "tools", "*.py")):
        for ref in re.findall(r"tools/([A-Za-z0-9_/]+\.(?:py|sh|hip))", open(f).read()):
            if not os.path.exists(os.path.join(root, "tools", ref)):
                missing.append((os.path.basename(f), ref))
    assert not missing, missing


def test_bench_valu_roofline_object():
    """bench.valu_roofline: the scans' binding ceiling in the bench line -- executed VALU wave-instructions per launch
    (counter profile) priced with the class mix of the chunk loop (static count) and the chip's issue prices, over the SIMDs."""
    import importlib.util
    import os
    from caduceus_amd import _build
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    pmc = {"sq": {"scan_bwd": {"SQ_INSTS_VALU": 1.2e9, "SQ_WAVES": 2048.0, "SQ_WAVE_CYCLES": 4.0e9, "SQ_ACTIVE_INST_VALU": 1.5e9,
                               "SQ_WAIT_INST_ANY": 8.0e8, "SQ_LDS_BANK_CONFLICT": 1.0e8, "SQ_LDS_IDX_ACTIVE": 5.0e8}}}
    isa = {"kernels": {"scan_bwd": {"chunk_loop_static": {"valu": 1000, "valu_pk": 800, "valu_dpp": 400, "trans": 200, "salu": 900, "lds": 150},
                                    "positions_per_chunk": 512, "vgprs": 249, "scratch_bytes": 0}}}
    v = bench.valu_roofline("scan_bwd", pmc, isa, 3.4, 1024)
    mean = (1000 * 1.15 + 800 * 1.95 + 400 * 1.9 + 200 * 3.4) / 2400  # SALU / LDS instructions are not VALU issue slots
    assert abs(v["mean_price_ns"] - mean) < 1e-9
    assert abs(v["issue_ceiling_ms"] - 1.2e9 * mean / 1024 * 1e-6) < 1e-9 and abs(v["kernel_over_issue_ceiling"] - 3.4 / v["issue_ceiling_ms"]) < 1e-9
    assert abs(v["valu_active_share_of_wave_cycles"] - 0.375) < 1e-12 and v["insts_valu_per_wave"] == 1.2e9 / 2048
    assert bench.valu_roofline("scan_fwd", pmc, isa, 1.0, 1024) is None and bench.valu_roofline("scan_bwd", None, isa, 1.0, 1024) is None
    # the committed static mix belongs to the scan sources in the tree (tools/make_scan_isa_json.py regenerates it)
    path = os.path.join(ROOT, bench.SCAN_ISA_FILE)
    if os.path.exists(path):
        committed = json.load(open(path))
        assert committed["scan_src"] == _build.scan_source_hash(), bench.SCAN_ISA_FILE + " was counted on other scan sources: re-run the tool"
        for kind in ("scan_fwd", "scan_bwd"):
            k = committed["kernels"][kind]
            assert k["scratch_bytes"] == 0 and k["vgprs"] <= 256 and k["valu_total"] > 1000
