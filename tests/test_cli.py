"""CLI surface of fit-gnn_amd/main.py vs the reference's main.py (flags, defaults, arg_correction), the Planetoid
raw-file loader (CPU tier; reference files are read only when /root/reference is present), and an end-to-end
GPU run of both the baseline and the FIT-GNN path on a synthetic Cora-shaped dataset."""
import ast
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "fit-gnn_amd"))
import main as cli  # noqa: E402

REF_MAIN = "/root/reference/main.py"
REF_CORA = "/root/reference/Baselines/SGGC/APPNP/dataset/cora/raw"


def test_arg_correction_rules():
    p = cli.build_parser()
    a = cli.arg_correction(p.parse_args(["--output_dir", "x"]))
    assert a.baseline is True and a.train_fitgnn is False          # main.py:125-128
    a = cli.arg_correction(p.parse_args(["--output_dir", "x", "--train_fitgnn", "--baseline"]))
    assert a.baseline is False
    a = cli.arg_correction(p.parse_args(["--output_dir", "x", "--cluster_node", "--extra_node"]))
    assert a.cluster_node and not a.extra_node                     # main.py:118-119
    a = cli.arg_correction(p.parse_args(["--output_dir", "x", "--dataset", "Physics"]))
    assert a.experiment == "random"                                # main.py:122-124


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference not mounted")
def test_flags_and_defaults_match_reference_main_py():
    src = open(REF_MAIN).read()
    ref = {}
    for m in re.finditer(r"parser\.add_argument\((.*)\)", src):
        call = ast.parse("f(" + m.group(1).split("#")[0].rstrip().rstrip(")") + ")").body[0].value
        name = call.args[0].value
        kw = {k.arg: k.value for k in call.keywords}
        default = ast.literal_eval(kw["default"]) if "default" in kw else (False if "action" in kw else None)
        ref[name] = (default, "action" in kw, "required" in kw)
    assert len(ref) == 33
    mine = {a.option_strings[0]: a for a in cli.build_parser()._actions if a.option_strings and a.option_strings[0] != "-h"}
    for name, (default, is_flag, required) in ref.items():
        assert name in mine, name
        assert mine[name].default == default, name
        assert (mine[name].nargs == 0) == is_flag, name
        assert mine[name].required == required, name


@pytest.mark.skipif(not os.path.exists("/root/reference/inference.py"), reason="reference not mounted")
def test_inference_flags_and_defaults_match_reference():
    import inference as icli

    src = open("/root/reference/inference.py").read()
    ref = {}
    for m in re.finditer(r"parser\.add_argument\((.*)\)", src):
        call = ast.parse("f(" + m.group(1).split("#")[0].rstrip().rstrip(")") + ")").body[0].value
        kw = {k.arg: k.value for k in call.keywords}
        ref[call.args[0].value] = (ast.literal_eval(kw["default"]) if "default" in kw else (False if "action" in kw else None),
                                   "action" in kw)
    assert len(ref) == 33
    mine = {a.option_strings[0]: a for a in icli.build_parser()._actions if a.option_strings and a.option_strings[0] != "-h"}
    for name, (default, is_flag) in ref.items():
        assert name in mine and mine[name].default == default and (mine[name].nargs == 0) == is_flag, name


@pytest.mark.skipif(not os.path.exists(REF_CORA), reason="reference not mounted")
def test_planetoid_loader_on_reference_cora_files():
    from fitgnn_amd import pipeline

    data, C = pipeline.load_planetoid(REF_CORA, "cora")
    assert (data.num_nodes, data.x.shape[1], C) == (2708, 1433, 7)          # dataset_info.csv:5
    assert data.edge_index.shape[1] == 2 * 5278
    assert int(data.train_mask.sum()) == 140 and int(data.val_mask.sum()) == 500 and int(data.test_mask.sum()) == 1000


def test_splits():
    from fitgnn_amd import pipeline

    d, C = pipeline.synthetic_dataset("synthetic-cora", seed=1)
    assert d.x.shape == (2708, 1433) and d.edge_index.shape == (2, 2 * 5278)
    d = pipeline.splits_classification(d, C, "random", np.random.default_rng(0))
    assert int(d.train_mask.sum()) == 20 * C and int(d.val_mask.sum()) == 30 * C
    assert not bool((d.train_mask & d.val_mask).any()) and not bool((d.train_mask & d.test_mask).any())
    assert int(d.train_mask.sum() + d.val_mask.sum() + d.test_mask.sum()) == 2708


@pytest.mark.gpu
def test_cli_end_to_end_on_synthetic_cora(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    common = ["--dataset", "synthetic-cora", "--runs", "1", "--hidden", "64", "--seed", "0", "--normalize_features"]
    _, acc, _ = cli.main(common + ["--output_dir", "b", "--baseline", "--epochs1", "40"])
    assert acc[0] > 0.5, acc  # 7 classes, homophilous synthetic labels: far above 1/7
    assert os.path.exists("save/node_cls/baseline/b/model.pt") and os.path.exists("results/baseline/synthetic-cora.csv")
    for setup, extra in (("Gs_train_2_Gs_infer", ["--cluster_node"]), ("Gs_train_2_Gs_infer", ["--extra_node", "--gradient_method", "MB", "--lr", "0.002"]),
                         ("Gs_train_2_Gs_infer", ["--extra_node", "--use_community_detection"]),
                         ("Gs_train_2_Gs_infer", ["--extra_node"]), ("Gc_train_2_Gs_infer", []), ("Gc_train_2_Gs_train", ["--extra_node"])):
        _, acc, _ = cli.main(common + ["--output_dir", "f", "--train_fitgnn", "--exp_setup", setup, "--coarsening_ratio", "0.5",
                                       "--epochs1", "40", "--epochs2", "40"] + extra)
        assert acc[0] > 0.4, (setup, acc)
    rows = open("results/synthetic-cora.csv").read().strip().split("\n")
    assert rows[0].startswith("dataset,coarsening_method,coarsening_ratio") and len(rows) == 7
    import inference as icli
    t_gs, acc_gs = icli.main(["--dataset", "synthetic-cora", "--hidden", "64", "--seed", "0", "--normalize_features", "--extra_node",
                              "--num_test_samples", "30", "--path_gs", "save/node_cls/f/", "--baseline",
                              "--path_b", "save/node_cls/baseline/b/", "--model_name_b", "model.pt"])
    assert acc_gs > 0.4 and t_gs < 0.05
    rows = open("inference_results/node_cls.csv").read().strip().split("\n")
    assert rows[0].startswith("dataset,baseline,experiment,exp_setup") and len(rows) == 3
    sd = torch.load("save/node_cls/f/model.pt")
    assert sorted(sd) == ["conv.0.bias", "conv.0.lin.weight", "conv.1.bias", "conv.1.lin.weight", "lt1.bias", "lt1.weight"]


@pytest.mark.gpu
def test_cli_trains_attention_layers_on_the_subgraph_union(tmp_path, monkeypatch):
    """`--layer_name GATConv` through the kept command line (network.py:13-17): Gs training in GD mode runs the first attention
    layer on the de-duplicated table and the last one aggregate-first on the loss rows (ops.FusedGATLastLayerRows); inference on the
    subgraphs with the saved weights; the state_dict carries PyG's GATConv keys."""
    monkeypatch.chdir(tmp_path)
    common = ["--dataset", "synthetic-cora", "--runs", "1", "--hidden", "64", "--seed", "0", "--normalize_features", "--layer_name", "GATConv"]
    _, acc, _ = cli.main(common + ["--output_dir", "g", "--train_fitgnn", "--exp_setup", "Gs_train_2_Gs_infer", "--coarsening_ratio", "0.5",
                                   "--epochs1", "40", "--epochs2", "60", "--extra_node"])
    assert acc[0] > 0.4, acc   # 7 classes, homophilous synthetic labels
    sd = torch.load("save/node_cls/g/model.pt")
    assert sorted(sd) == ["conv.0.att_dst", "conv.0.att_src", "conv.0.bias", "conv.0.lin.weight", "conv.1.att_dst", "conv.1.att_src",
                          "conv.1.bias", "conv.1.lin.weight", "lt1.bias", "lt1.weight"]


@pytest.mark.gpu
def test_cli_graph_regression_on_synthetic_qm9(tmp_path, monkeypatch):
    """main.py on the QM9-shaped stand-in (BASELINE.json config 5's plumbing): batched coarsening of every molecule,
    Gs / Gc training loops, results row with the reference's columns; the model must beat predicting zero."""
    monkeypatch.chdir(tmp_path)
    common = ["--dataset", "synthetic-qm9", "--n_graphs", "600", "--hidden", "64", "--seed", "0", "--train_fitgnn", "--batch_size", "64",
              "--lr", "0.002", "--property", "0", "--epochs1", "15", "--epochs2", "15", "--output_dir", "q"]
    losses = {}
    for setup, extra in (("Gs_train_2_Gs_infer", ["--extra_node"]), ("Gc_train_2_Gc_infer", []), ("Gc_train_2_Gs_train", ["--cluster_node"])):
        losses[setup] = cli.main(common + ["--exp_setup", setup] + extra)
        assert np.isfinite(losses[setup])
    rows = open("results/synthetic-qm9.csv").read().strip().split("\n")
    assert rows[0].startswith("dataset,coarsening_method,coarsening_ratio,exp_setup") and rows[0].endswith("property_idx}") and len(rows) == 4
    assert os.path.exists("save/graph_reg/q/model.pt")


@pytest.mark.gpu
def test_cli_node_regression_on_synthetic_chameleon(tmp_path, monkeypatch):
    """main.py node regression (run.py:508-573) on a chameleon-shaped stand-in: Regress_node on Gs, L1 / std(labels);
    a trained model must beat the constant predictor (normalised L1 of the mean predictor is ~0.8)."""
    monkeypatch.chdir(tmp_path)
    losses, _ = cli.main(["--dataset", "synthetic-chameleon", "--runs", "1", "--hidden", "64", "--seed", "0", "--train_fitgnn", "--extra_node",
                          "--exp_setup", "Gs_train_2_Gs_infer", "--epochs2", "150", "--lr", "0.01", "--output_dir", "r", "--coarsening_ratio", "0.5"])
    assert losses[0] < 0.76, losses  # constant predictor: ~0.80
    rows = open("results/synthetic-chameleon.csv").read().strip().split("\n")
    assert rows[0].startswith("dataset,coarsening_method,coarsening_ratio,layer_name") and len(rows) == 2
    import inference as icli
    t, _ = icli.main(["--dataset", "synthetic-chameleon", "--hidden", "64", "--seed", "0", "--extra_node", "--num_test_samples", "20",
                      "--path_gs", "save/node_reg/r/", "--coarsening_ratio", "0.5"])
    assert t < 0.05 and os.path.exists("inference_results/node_reg.csv")


@pytest.mark.gpu
def test_cli_graph_classification_on_synthetic_proteins(tmp_path, monkeypatch):
    """main.py graph classification (run.py:575-706) on a PROTEINS-shaped stand-in: Classify_graph_gc / _gs with the
    reference's softmax + CrossEntropyLoss, results row with best_test_acc; both classes are separable by structure and
    node labels, so accuracy must clear chance."""
    monkeypatch.chdir(tmp_path)
    common = ["--dataset", "synthetic-proteins", "--n_graphs", "400", "--hidden", "64", "--seed", "0", "--train_fitgnn", "--batch_size", "50",
              "--lr", "0.005", "--epochs1", "30", "--epochs2", "30", "--output_dir", "p"]
    for setup in ("Gs_train_2_Gs_infer", "Gc_train_2_Gc_infer"):
        loss, acc = cli.main(common + ["--exp_setup", setup])
        assert np.isfinite(loss) and acc > 0.6, (setup, loss, acc)
    rows = open("results/synthetic-proteins.csv").read().strip().split("\n")
    assert rows[0].endswith("best_test_loss,best_test_acc") and len(rows) == 3
    # inference.py on the last checkpoint (Gc_train_2_Gc_infer wrote save/graph_cls/p/model.pt)
    import inference as icli
    t, acc = icli.main(["--dataset", "synthetic-proteins", "--n_graphs", "400", "--hidden", "64", "--seed", "0", "--num_test_samples", "25",
                        "--exp_setup", "Gc_train_2_Gc_infer", "--path_gc", "save/graph_cls/p/", "--model_name_gc", "model.pt"])
    assert acc > 0.6 and t < 0.05
    assert os.path.exists("inference_results/graph_cls.csv")


@pytest.mark.gpu
def test_cli_baselines_of_the_other_tasks(tmp_path, monkeypatch):
    """--baseline for node regression and the graph-level tasks (run.py:904-1100): full-graph / uncoarsened-graph models,
    results under results/baseline/."""
    monkeypatch.chdir(tmp_path)
    losses, _ = cli.main(["--dataset", "synthetic-chameleon", "--runs", "1", "--hidden", "64", "--seed", "0", "--baseline", "--epochs1", "200",
                          "--output_dir", "b"])
    assert losses[0] < 0.79, losses   # constant predictor ~0.80; 683 train nodes, dropout 0.5
    loss = cli.main(["--dataset", "synthetic-qm9", "--n_graphs", "400", "--hidden", "64", "--seed", "0", "--baseline", "--batch_size", "50",
                     "--lr", "0.002", "--epochs1", "15", "--output_dir", "b"])
    assert np.isfinite(loss)
    loss, acc = cli.main(["--dataset", "synthetic-proteins", "--n_graphs", "400", "--hidden", "64", "--seed", "0", "--baseline",
                          "--batch_size", "50", "--lr", "0.005", "--epochs1", "30", "--output_dir", "b"])
    assert np.isfinite(loss) and acc > 0.6
    for name in ("synthetic-chameleon", "synthetic-qm9", "synthetic-proteins"):
        assert os.path.exists(f"results/baseline/{name}.csv")
