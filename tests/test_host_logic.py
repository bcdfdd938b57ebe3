"""CPU: host-side logic of the boundary (model recognition, sharding, failure without a GPU)."""
import numpy as np
import pytest
import torch

from oracle import fields as ofields, synth


def test_specs_agree_with_oracle():
    from mirender import fields
    for kind, name in fields.KIND_NAMES.items():
        assert [(k, tuple(s)) for k, s in fields.SPECS[kind]] == [(k, tuple(s)) for k, s in ofields.SPECS[name]]
        assert fields.MACS[kind] == ofields.MACS[name]
    assert fields.FLOPS_PER_POINT[fields.NERF] == 1182976
    assert fields.FLOPS_PER_POINT[fields.SIREN_NERF] == 1119232
    assert fields.FLOPS_PER_POINT[fields.FILM_SIREN_NERF] == 1053696


@pytest.mark.parametrize("name", ["nerf", "siren_nerf", "film_siren_nerf", "film_siren_nerf_nodir", "tiny_nerf"])
def test_detect_kind_by_layout(name):
    from mirender import fields
    sd = synth.state_dict(name, seed=0)
    assert fields.KIND_NAMES[fields.detect_kind(sd)] == name
    bad = dict(sd)
    bad.pop(next(iter(bad)))
    assert fields.detect_kind(bad) is None
    wrong = {k: (v if i else torch.zeros(7, 7)) for i, (k, v) in enumerate(sd.items())}
    assert fields.detect_kind(wrong) is None


def test_module_families_have_reference_keys_and_init():
    from mirender import fields
    torch.manual_seed(0)
    for cls, name in ((fields.NeRF, "nerf"), (fields.SirenNeRF, "siren_nerf"), (fields.TinyNeRF, "tiny_nerf"),
                      (fields.FilmSirenNeRF, "film_siren_nerf")):
        m = cls()
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == \
            {k: tuple(v) for k, v in ofields.param_shapes(name).items()}
        m.load_state_dict(synth.state_dict(name, seed=1))          # reference-layout checkpoints load
    m = fields.FilmSirenNeRF(use_dir=False)
    assert tuple(m.hidden_layer_rgb.weight.shape) == (256, 256) and m.use_dir is False
    # initialiser ranges (nerf/nerf.py:25-28,114-117,134; pi_GAN/modules.py:27-31)
    n = fields.NeRF()
    assert float(n.layers_pos[0].weight.abs().max()) <= np.sqrt(2) * np.sqrt(6 / (60 + 256)) + 1e-6
    assert float(n.layers_pos[0].bias.abs().max()) == 0
    s = fields.SirenNeRF()
    assert float(s.layers_pos[0].weight.abs().max()) <= 1 / 30 + 1e-7
    assert float(s.layers_pos[1].weight.abs().max()) <= np.sqrt(6 / 256) / 30 + 1e-7
    f = fields.FilmSirenNeRF()
    assert float(f.input_layer.weight.abs().max()) <= 1 / 3 + 1e-7


def test_no_cpu_fallback():
    """Without a ROCm device the product refuses to run (it must never fall back to PyTorch/CPU)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mirender import _lib, fields, render_core
    m = fields.NeRF()
    with pytest.raises(_lib.MiRenderError):
        render_core.render_rays(torch.zeros(4, 2, 3), 2.0, 6.0, m, m, 8, 8)
    with pytest.raises(_lib.MiRenderError):
        m(torch.zeros(4, 6))
    with pytest.raises((_lib.MiRenderError, RuntimeError)):
        render_core.get_rays(4, 4, 5.0, np.eye(4, dtype=np.float32))


def test_shard_ranges_partition():
    from mirender.dist import shard_range
    for total in (640000, 160000, 10, 7, 1):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_film_table_from_reference_style_params():
    from mirender import fields
    m = fields.FilmSirenNeRF()
    with pytest.raises(ValueError):
        fields.film_table(m)
    mapping = synth.film_params(1, seed=3)[0]
    m.set_film_params(mapping)                       # list of 9 (gamma, beta) chunks, pi_GAN/modules.py:96-99
    assert len(m.film_params) == 9 and m.film_params[0][0].shape == (256,)
    assert torch.equal(fields.film_table(m)[0], mapping)


def test_pigan_generator_state_dict_layout():
    """Generator's parameters carry the reference's names (pi_GAN/modules.py:165-175) so its checkpoints
    ('generator' entry of pi_GAN/train.py:162-172) load."""
    from mirender import pigan
    g = pigan.Generator(64, 16)
    keys = set(g.state_dict().keys())
    assert {"film_siren_nerf.input_layer.weight", "film_siren_nerf.hidden_layers.6.bias",
            "film_siren_nerf.output_layer_sigma.0.weight", "film_siren_nerf.hidden_layer_rgb.weight",
            "film_siren_nerf.output_layer_rgb.0.bias", "mapping_network.input_layer.0.weight",
            "mapping_network.hidden_layers.0.weight", "mapping_network.hidden_layers.2.bias",
            "mapping_network.output_layers.8.weight"} <= keys
    assert sum(p.numel() for p in g.parameters()) == 529156 + (64 * 256 + 256) + 2 * (256 * 256 + 256) + 9 * (256 * 512 + 512)
    out = g.get_mapping(torch.zeros(2, 64))
    assert tuple(out.shape) == (2, 9, 512)
    r = g.renderer
    assert abs(float(r.focal) - 16 / 2 / np.tan(6 * np.pi / 180)) < 1e-9
    g.set_resolution(32)
    assert r.width == 32 and abs(float(r.focal) - 32 / 2 / np.tan(6 * np.pi / 180)) < 1e-9


def test_reference_class_layouts_are_recognised():
    """tests/golden/ref_layouts.json = {name: shape} of named_parameters() of the REFERENCE's own classes
    (nerf/nerf.py NeRF, SirenNeRF; pi_GAN/modules.py FilmSirenNeRF(use_dir=+-), Generator, MappingNetwork), dumped by
    tests/golden/make_golden.py in the build container.  detect_kind must recognise each field class, and our
    Generator / MappingNetwork must carry exactly the reference's state-dict keys and shapes."""
    import json
    import os
    from mirender import fields, pigan
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_layouts.json")) as f:
        ref = json.load(f)
    expect = {"nerf.NeRF": fields.NERF, "nerf.SirenNeRF": fields.SIREN_NERF,
              "pi_GAN.FilmSirenNeRF(use_dir=True)": fields.FILM_SIREN_NERF,
              "pi_GAN.FilmSirenNeRF(use_dir=False)": fields.FILM_SIREN_NERF_NODIR}
    for name, kind in expect.items():
        named = {k: torch.empty(s) for k, s in ref[name].items()}
        assert fields.detect_kind(named) == kind, name
        ours = {fields.NERF: fields.NeRF, fields.SIREN_NERF: fields.SirenNeRF, fields.FILM_SIREN_NERF: fields.FilmSirenNeRF,
                fields.FILM_SIREN_NERF_NODIR: lambda: fields.FilmSirenNeRF(use_dir=False)}[kind]()
        assert {k: list(v.shape) for k, v in ours.named_parameters()} == ref[name]
        assert list(dict(ours.named_parameters())) == list(ref[name]) or sorted(dict(ours.named_parameters())) == sorted(ref[name])
    g = pigan.Generator(256, 64)
    assert {k: list(v.shape) for k, v in g.state_dict().items()} == ref["pi_GAN.Generator(256, 64).state_dict"]
    assert {k: list(v.shape) for k, v in g.named_parameters()} == ref["pi_GAN.Generator(256, 64)"]
    assert {k: list(v.shape) for k, v in pigan.MappingNetwork().named_parameters()} == ref["pi_GAN.MappingNetwork()"]
    # the field inside a reference Generator is found under its attribute name
    inner = {k[len("film_siren_nerf."):]: torch.empty(s) for k, s in ref["pi_GAN.Generator(256, 64)"].items()
             if k.startswith("film_siren_nerf.")}
    assert fields.detect_kind(inner) == fields.FILM_SIREN_NERF


def test_bench_metric_is_baseline_jsons_metric():
    """bench.py's headline line carries BASELINE.json's metric string verbatim (the driver matches on it)."""
    import json
    import os
    import bench
    with open(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "BASELINE.json")) as f:
        assert bench.METRIC == json.load(f)["metric"]


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` from a plain shell (no torch.distributed.run environment) must start N fresh rank
    processes itself - before any GPU call, never by re-exec'ing - and relay their exit code."""
    import os
    import sys
    import bench
    cmd = bench.spawn_command(["--gpus", "4", "--steps", "3", "--warmup", "1"], 4, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    seen = {}

    class Done:
        returncode = 7

    def fake_run(c, env=None, **kw):
        seen["cmd"], seen["env"] = c, env
        return Done()
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "2"])
    monkeypatch.setattr(torch.cuda, "set_device", lambda *_: (_ for _ in ()).throw(AssertionError("parent touched the GPU")))
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-4:] == ["--gpus", "2", "--steps", "2"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def _rebuild_from_attrs(kind_name, layers):
    """A torch module tree whose layer objects carry exactly what tests/golden/ref_layer_attrs.json recorded from the
    reference's own classes (class name, activation_name, w_0, the module after a Linear in its Sequential)."""
    from mirender import fields
    kind = {v: k for k, v in fields.KIND_NAMES.items()}[kind_name]
    root = torch.nn.Module()
    for key, (o, i) in fields.SPECS[kind]:
        rec = layers[key]
        if rec["class"] == "Linear":
            leaf = torch.nn.Linear(i, o)
        else:
            leaf = type(rec["class"], (torch.nn.Linear,), {})(i, o)
        for attr in ("activation_name", "w_0"):
            if attr in rec:
                setattr(leaf, attr, rec[attr])
        parts = key.split(".")
        if len(parts) == 1:
            root.add_module(parts[0], leaf)
        elif "next_in_sequential" in rec:
            root.add_module(parts[0], torch.nn.Sequential(leaf, getattr(torch.nn, rec["next_in_sequential"])()))
        else:
            if not hasattr(root, parts[0]):
                root.add_module(parts[0], torch.nn.ModuleList())
            getattr(root, parts[0]).append(leaf)
    return kind, root


def test_same_layout_other_arithmetic_is_not_claimed():
    """a recognised layout is not enough - the kernels hard-code each layer's activation.  The reference's
    own classes (their layer objects' self-description dumped into ref_layer_attrs.json by importing them) must pass -
    FilmSirenNeRF(w_0=25) (pi_GAN/modules.py:73) included since round 4: the FiLM kernels take the module's w_0 at run time -
    while FiLM layers that disagree about w_0, a Dense stack with another activation name (nerf/nerf.py:15-16) and a bare
    nn.Linear stack must fall to the generic path (as_packed_field -> None)."""
    import json
    import os
    from mirender import fields
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_layer_attrs.json")) as f:
        ref = json.load(f)
    for name, rec in ref.items():
        kind, m = _rebuild_from_attrs(rec["kind"], rec["layers"])
        assert fields.detect_kind(dict(m.named_parameters())) == kind, name
        why = fields.hyper_mismatch(m, kind)
        assert why is None, (name, why)
        assert fields.film_w0(m, kind) == (25.0 if "w_0=25" in name else 30.0), name
    # one FiLM layer with another w_0 than the rest: one scalar per network is all the kernels take
    layers = json.loads(json.dumps(ref["pi_GAN.FilmSirenNeRF(w_0=25)"]["layers"]))
    layers["hidden_layers.3"]["w_0"] = 30
    kind, m = _rebuild_from_attrs("film_siren_nerf", layers)
    why = fields.hyper_mismatch(m, kind)
    assert why is not None and "w_0" in why and fields.as_packed_field(m) is None    # generic path; no device needed to decide
    layers["hidden_layers.3"]["w_0"] = -25
    kind, m = _rebuild_from_attrs("film_siren_nerf", layers)
    assert "w_0" in fields.hyper_mismatch(m, kind)
    # NeRF layout, one hidden Dense switched to tanh; SirenNeRF layout whose sin layers are Dense('relu')
    layers = json.loads(json.dumps(ref["nerf.NeRF"]["layers"]))
    layers["layers_pos.3"]["activation_name"] = "tanh"
    kind, m = _rebuild_from_attrs("nerf", layers)
    assert "tanh" in fields.hyper_mismatch(m, kind) and fields.as_packed_field(m) is None
    layers = json.loads(json.dumps(ref["nerf.SirenNeRF"]["layers"]))
    for k in layers:
        if layers[k]["class"] == "Siren":
            layers[k] = {"class": "Dense", "activation_name": "relu"}
    kind, m = _rebuild_from_attrs("siren_nerf", layers)
    assert fields.hyper_mismatch(m, kind) is not None and fields.as_packed_field(m) is None
    # a stack of bare nn.Linear says nothing about its activations: not claimed either
    layers = {k: {"class": "Linear"} for k in ref["nerf.NeRF"]["layers"]}
    kind, m = _rebuild_from_attrs("nerf", layers)
    assert fields.hyper_mismatch(m, kind) is not None
    # our own modules name their activations and are claimed
    for cls in (fields.NeRF, fields.TinyNeRF, fields.SirenNeRF, fields.FilmSirenNeRF, fields.FilmSirenNeRFNoDir):
        assert fields.hyper_mismatch(cls(), cls.KIND) is None, cls
    # ... and keep the reference's constructor (pi_GAN/modules.py:73): w_0 / c shape the initialisation and w_0 the kernels
    m = fields.FilmSirenNeRF(w_0=25, c=4)
    assert fields.film_w0(m, m.KIND) == 25.0 and fields.hyper_mismatch(m, m.KIND) is None
    assert float(m.hidden_layers[2].weight.abs().max()) <= np.sqrt(4 / 256) / 25 + 1e-7
    assert isinstance(fields.FilmSirenNeRF(256, 8, 6, 30, False), fields.FilmSirenNeRFNoDir)
    with pytest.raises(fields._lib.MiRenderError):
        fields.FilmSirenNeRF(hidden_dim=128)


@pytest.mark.skipif(torch.cuda.is_available(), reason="uses the absence of a GPU as the rank failure")
@pytest.mark.timeout(300)
def test_bench_rank_failure_is_relayed_with_a_nonzero_exit_code():
    """`python bench.py --gpus 2` from a plain shell, on a host where the ranks cannot work (no GPU here): the spawned
    ranks fail, each failing rank's traceback reaches stderr tagged with its rank, no JSON line is printed and the exit
    code is non-zero - end to end through the real torch.distributed.run launcher."""
    import os
    import subprocess
    import sys
    import bench
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.abspath(bench.__file__), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode != 0
    assert "[bench.py rank 0] FAILED" in r.stderr or "[bench.py rank 1] FAILED" in r.stderr, r.stderr[-2000:]
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())
