"""runner.py: the reference's command-line surface (runner.py:9-65) — override parsing on the CPU, a short run on the GPU."""
import os

import pytest


def test_override_parsing_matches_the_reference_cli_shapes():
    import runner

    flat = runner.parse_overrides(["testlist=scan24,scan37", "vol=dtu_pn", "opt_stepNs=[30,0,0]", "vol.train.num_pixels=256", "grad_clip=false",
                                   "exps_folder=exps_x"])
    assert runner.scenes_of(flat) == ["scan24", "scan37"]
    assert flat["opt_stepNs"] == [30, 0, 0] and flat["grad_clip"] is False
    args = runner.nest(flat)
    assert args["exps_folder"] == "exps_x" and args["vol"]["train"]["num_pixels"] == 256
    assert args["vol"]["loss"]["tv_weight"] == 0.01 and args["vol"]["train"]["expname"] == "ours"     # config/ours.yaml defaults
    assert args.vol.get_int("train.checkpoint_freq") == 15000
    with pytest.raises(SystemExit):
        runner.parse_overrides(["scan24"])


@pytest.mark.gpu
def test_short_run_through_the_cli(tmp_path, capsys):
    import runner

    (vo,) = runner.main(["testlist=scan24", "vol=dtu_pn", "opt_stepNs=[6,0,0]", "points=1500", "vol.train.num_pixels=64",
                         "vol.train.checkpoint_freq=2", f"root={tmp_path}", "exps_folder=exps"])
    assert vo.iter_step == 6
    assert "finished training scan24" in capsys.readouterr().out
    assert os.path.exists(os.path.join(vo.checkpoints_path, "ModelParameters", "latest.pth"))
    assert os.path.exists(os.path.join(vo.checkpoints_path, "OptimizerParameters", "latest.pth"))
