"""Command-line entry with the reference's surface (runner.py:9-65): hydra-style `key=value` overrides of config/ours.yaml's
keys — `testlist= vol= outdir= exps_folder= opt_stepNs= grad_clip= is_continue=` and dotted `vol.train.num_pixels=...` — then,
per scene, `VolOpt(args, batch_size=1, is_continue, timestamp='latest', checkpoint='latest', scan)`, `gen_dataset(0)`,
`run(opt_stepNs[0])`; `mesh_resolution=N` (own addition) then sweeps `get_sdf_eval` over an N-cell grid and writes the iso-surface as `.ply`.

hydra / omegaconf are not needed (a literal `key=value` parser covers what the reference's own command lines use,
readme.md:65,85,88).  The reference's datasets are a separate download; `data=synthetic` (default) optimises the built-in
DTU-shaped synthetic scene instead, `points=` / `seed=` pick its size, `prior=fitted|kaiming` its local prior (fitted: the SDF is the
signed distance to the analytic surface), `local=true` adds synthetic `local_data` so that the feature-consistency term (weight 0.5) runs.  The optimisation itself is `spurfies_amd.train.VolOpt`
on the HIP path; `sync_free=true` (default) is the mode `bench.py` measures; `use_graph=true` replays forward + loss + backward as one
hipGraph, which takes the host's ~3.5 ms of launch work per step out of the loop (the loop, unlike the bench, also pays the reference's
per-step `randperm` over the image's pixels on the host).

    python runner.py testlist=scan24 vol=dtu_pn opt_stepNs=[200,0,0] exps_folder=exps_vsdf
"""
from __future__ import annotations

import ast
import sys
import time

DEFAULTS = {   # config/base.yaml + config/ours.yaml (the keys the optimisation path reads)
    "testlist": "scan24", "vol": "dtu_pn", "outdir": "exps_mvs", "exps_folder": "exps_vsdf", "opt_stepNs": [100000, 0, 0], "grad_clip": True,
    "is_continue": False, "data": "synthetic", "points": 10000, "seed": 0, "prior": "fitted", "local": False, "sync_free": True, "use_graph": False, "root": "./", "mesh_resolution": 0, "mesh_level": 0.0,
    "vol.train.expname": "ours", "vol.train.render_freq": 500, "vol.train.checkpoint_freq": 15000, "vol.train.num_pixels": 1024,
    "vol.train.split_n_pixels": 500, "vol.loss.local_weight": 0.5, "vol.loss.pseudo_weight": 0.5, "vol.loss.eikonal_weight": 0.001,
    "vol.loss.rgb_weight": 1.0, "vol.loss.tv_weight": 0.01, "vol.dataset.data_dir": "dtu",
}


def _literal(text: str):
    low = text.strip().lower()
    if low in ("true", "false"):
        return low == "true"
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def parse_overrides(argv) -> dict:
    """['testlist=scan24,scan37', 'opt_stepNs=[10,0,0]', 'vol.train.num_pixels=512'] -> flat dict over DEFAULTS."""
    flat = dict(DEFAULTS)
    for tok in argv:
        if "=" not in tok:
            raise SystemExit(f"runner.py: expected key=value, got {tok!r}")
        key, val = tok.split("=", 1)
        key = key.lstrip("+")
        flat[key] = _literal(val)
    return flat


def nest(flat: dict):
    """dotted keys -> nested spurfies_amd.conf.Conf (what VolOpt reads as `args` / `args['vol']`)."""
    from spurfies_amd.conf import Conf

    root: dict = {}
    for key, val in flat.items():
        node = root
        parts = key.split(".")
        if parts[0] == "vol" and len(parts) == 1:      # `vol=dtu_pn` names the config group; its content lives under vol.*
            root.setdefault("vol_name", val)
            continue
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = val

    def conv(d):
        return Conf({k: conv(v) if isinstance(v, dict) else v for k, v in d.items()})

    return conv(root)


def scenes_of(flat: dict):
    t = str(flat["testlist"])
    if t.endswith(".txt"):
        return [line.strip() for line in open(t) if line.strip()]
    return [x for x in t.replace(" ", "").split(",") if x]


def optimise_scene(scene_name: str, flat: dict, args):
    import numpy as np
    import torch

    from spurfies_amd import synthetic as syn
    from spurfies_amd.train import VolOpt

    if flat["data"] != "synthetic":
        raise SystemExit("runner.py: the reference's DTU / MipNeRF-360 loaders need its data download (out of scope, DESIGN.md §8); "
                         "use data=synthetic, or construct spurfies_amd.train.VolOpt(dataset=..., neural_points=...) from Python")
    from spurfies_amd.train import SyntheticDataset

    scene = syn.make_scene(int(flat["points"]), seed=int(flat["seed"]), prior=str(flat["prior"]))
    prior = {k: torch.from_numpy(np.asarray(v)) for k, v in scene["state"].items() if k.startswith(("F_geometry", "T."))}
    vol_opt = VolOpt(args=args, batch_size=1, is_continue=bool(flat["is_continue"]), timestamp="latest", checkpoint="latest", scan=scene_name,
                     root=str(flat["root"]), scene=scene, neural_points={"pts": scene["state"]["neural_pts"], "colors": scene["colors"]},
                     prior_state_dict=prior, device="cuda", sync_free=bool(flat["sync_free"]), use_graph=bool(flat["use_graph"]),
                     dataset=SyntheticDataset(scene, local=True) if bool(flat["local"]) else None,
                     # the fitted prior pairs with latents that carry the normals (synthetic.make_scene): START values — VolOpt applies them
                     # before it restores a checkpoint, so a resumed run (is_continue=true) keeps its trained latents
                     init_state_dict=({"neural_feats_geometry": torch.from_numpy(scene["state"]["neural_feats_geometry"])}
                                      if str(flat["prior"]) == "fitted" else None))
    vol_opt.gen_dataset(0)
    vol_opt.stg = 0
    steps = flat["opt_stepNs"]
    steps = steps[0] if isinstance(steps, (list, tuple)) else int(steps)
    t0 = time.perf_counter()
    epoch = vol_opt.run(int(steps))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    last = {k: float(v.detach()) for k, v in (vol_opt.last_losses or {}).items()}
    print(f"finished training {scene_name}: {vol_opt.iter_step} steps, epoch {epoch}, {1e3 * dt / max(vol_opt.iter_step, 1):.2f} ms/step, "
          f"loss {last.get('loss', float('nan')):.5f}, checkpoints in {vol_opt.checkpoints_path}")
    if int(flat["mesh_resolution"]) > 0:       # the reference's mesh route (utils/plots.py:188-333): grid, get_sdf_eval sweep, iso-surface
        import os

        from spurfies_amd.utils import surface

        model = vol_opt.model
        model.eval()
        pts = scene["state"]["neural_pts"]
        grid = surface.get_grid(pts, int(flat["mesh_resolution"]))
        vol = surface.sdf_volume(model.get_sdf_eval, grid)
        level = flat["mesh_level"]      # a number (the reference: 0), or "median": the synthetic scene's random prior has no zero level set
        level = float(np.median(vol[vol != surface.SDF_FILL])) if str(level) == "median" else float(level)
        verts, faces = surface.triangulate(vol, grid, level=level)
        verts, faces = surface.largest_component(verts, faces)          # plots.py:213-215
        path = os.path.join(os.path.dirname(vol_opt.checkpoints_path), f"surface_{vol_opt.iter_step}.ply")
        surface.write_ply(path, verts, faces)
        print(f"mesh: {grid['grid_points'].shape[0]} grid points, level {level:.4f}, {len(verts)} vertices, {len(faces)} faces -> {path}")
        if len(faces):      # the synthetic scene has a ground truth: the analytic surface (evals/eval_dtu.py-style accuracy / completeness)
            rng = np.random.default_rng(0)
            d = rng.standard_normal((100000, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            gt = d * syn.lobed_radius(d, scene["base_radius"])[:, None]
            h = float(grid["xyz"][0][1] - grid["xyz"][0][0])
            res = surface.chamfer_dtu(surface.sample_mesh_points(verts, faces, 0.5 * h), gt, max_dist=20 * h, thresh=0.5 * h)
            print(f"chamfer vs the analytic surface (grid step {h:.4f}): accuracy {res['accuracy']:.5f}, completeness {res['completeness']:.5f}, "
                  f"overall {res['overall']:.5f}")
    return vol_opt


def main(argv=None):
    flat = parse_overrides(sys.argv[1:] if argv is None else argv)
    args = nest(flat)
    out = []
    for scene in scenes_of(flat):
        out.append(optimise_scene(scene, flat, args))
    return out


if __name__ == "__main__":
    main()
