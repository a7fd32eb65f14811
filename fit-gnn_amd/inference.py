#!/usr/bin/env python3
"""inference.py -- FIT-GNN's per-query inference timing command line on the MI355X implementation.

Same flags and defaults as the reference's inference.py:221-254 (note its own defaults differ from main.py's:
--lr 0.001, no --layer_name), same arg_correction, same CSV row appended to inference_results/<task>.csv
(inference.py:826-874).  Node classification (per-query: the one subgraph holding the node) and the graph-level tasks (per sampled graph: its
subgraph set or coarsened graph).  For every sampled test node the model runs on
the ONE subgraph that contains it (inference.py:668-688) -- that is the "inference that FITs in memory" claim --
and, with --baseline, on the full graph (inference.py:651-666).  Unlike the reference, the timed region is
bracketed by a device synchronisation (the reference's time() around an asynchronous launch measures launch time).
Extra flags: --data_root, --device, --layer_name (the reference hard-codes GCN in its Net1, inference.py:22-50).
"""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import main as train_cli  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, default='cora')
    p.add_argument('--experiment', type=str, default='fixed')
    p.add_argument('--runs', type=int, default=20)
    p.add_argument('--exp_setup', type=str, default='Gc_train_2_Gs_infer')
    p.add_argument('--hidden', type=int, default=512)
    p.add_argument('--epochs1', type=int, default=100)
    p.add_argument('--epochs2', type=int, default=300)
    p.add_argument('--num_layers1', type=int, default=2)
    p.add_argument('--num_layers2', type=int, default=2)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--train_ratio', type=float, default=0.3)
    p.add_argument('--val_ratio', type=float, default=0.2)
    p.add_argument('--early_stopping', type=int, default=10)
    p.add_argument('--extra_node', action='store_true')
    p.add_argument('--cluster_node', action='store_true')
    p.add_argument('--lr', type=float, default=0.001)
    p.add_argument('--weight_decay', type=float, default=0.0005)
    p.add_argument('--use_community_detection', action='store_true')
    p.add_argument('--normalize_features', action='store_true')
    p.add_argument('--coarsening_ratio', type=float, default=0.5)
    p.add_argument('--coarsening_method', type=str, default='variation_neighborhoods')
    p.add_argument('--task', type=str, default='node_cls')
    p.add_argument('--seed', type=int, default=None)
    p.add_argument('--multi_prop', action='store_true')
    p.add_argument('--property', type=int, default=0)
    p.add_argument('--num_test_samples', type=int, default=20)
    p.add_argument('--path_b', type=str, default="./save/node_cls/baseline/")
    p.add_argument('--model_name_b', type=str, default="baseline_cora_fixed.pt")
    p.add_argument('--path_gs', type=str, default="./save/node_cls/cora_fixed_Gc_train_2_Gs_infer_0.5_variation_neighborhoods_cluster/")
    p.add_argument('--model_name_gs', type=str, default="model.pt")
    p.add_argument('--path_gc', type=str, default="./save/node_cls/cora_Gc_train_2_Gc_infer_0.5_variation_neighborhoods_extra/")
    p.add_argument('--model_name_gc', type=str, default="model.pt")
    p.add_argument('--baseline', action='store_true')
    # not in the reference
    p.add_argument('--data_root', type=str, default='./dataset')
    p.add_argument('--device', type=str, default='cuda')
    p.add_argument('--layer_name', type=str, default='GCNConv')
    p.add_argument('--n_graphs', type=int, default=2000)
    p.add_argument('--community_nodes', type=int, default=165000)
    return p


def arg_correction(args):
    """inference.py:118-124 (no train_fitgnn flag here: --baseline is taken as given)."""
    if args.cluster_node:
        args.extra_node = False
    elif args.extra_node:
        args.cluster_node = False
    if args.experiment == 'fixed' and args.dataset in ('ogbn-products', 'dblp', 'Physics', 'WikiCS', 'Flickr'):
        args.experiment = 'random'
    return args


def timed_forward(model, x, ei, device):
    torch.cuda.synchronize(device)
    t0 = time.time()
    out = model(x, ei)
    torch.cuda.synchronize(device)
    return out, time.time() - t0


def graph_inference(args, mol):
    """inference.py:288-538 (graph_cls / graph_reg): per sampled dataset graph, one timed forward of the FIT-GNN model on
    the graph's subgraph set (or its coarsened graph for Gc_train_2_Gc_infer) and, with --baseline, of the baseline on
    the uncoarsened graph; accuracy (graph_cls) or L1 loss (graph_reg) over the samples; one CSV row per model."""
    import types

    from fitgnn_amd import graph_data, network
    from fitgnn_amd.train import _cat_pieces

    dev = torch.device(args.device)
    cls_task = args.task == "graph_cls"
    gset = graph_data.GraphSet(mol, ratio=args.coarsening_ratio, extra_node=bool(args.extra_node), device=dev,
                               cluster_node=bool(args.cluster_node))
    rng = np.random.default_rng(args.seed)
    ids = rng.choice(gset.n_graphs, size=min(args.num_test_samples, gset.n_graphs), replace=False).tolist()
    use_gc = args.exp_setup == "Gc_train_2_Gc_infer"
    kind = "gc" if use_gc else "gs"
    args.num_layers1 = args.num_layers2
    if not cls_task:
        args.num_classes = 1
    Model = {(True, True): network.Classify_graph_gc, (True, False): network.Classify_graph_gs,
             (False, True): network.Regress_graph_gc, (False, False): network.Regress_graph_gs}[(cls_task, use_gc)]

    def run(model, kind):
        model.eval()
        times, losses, hits = [], [], 0
        with torch.no_grad():
            for g in ids:
                b = _cat_pieces([gset.batch(g, g + 1, kind)], kind, types)
                torch.cuda.synchronize(dev)
                t0 = time.time()
                out = model(b, b["graph_of_masked"]) if kind == "gs" else model(b["gc"])
                torch.cuda.synchronize(dev)
                times.append(time.time() - t0)
                y = b["y"].long()
                if cls_task:
                    losses.append(float(F.cross_entropy(out, y.flatten())))
                    hits += int(out.argmax(1) == y.flatten())
                else:
                    losses.append(float(F.l1_loss(out, y[:, args.property].view(-1, 1).float())))
        t = float(np.mean(times[1:])) if len(times) > 1 else float(times[0])
        return t, float(np.mean(losses)), hits / len(ids)

    model = Model(args).to(dev)
    model.load_state_dict(torch.load(os.path.join(args.path_gc if use_gc else args.path_gs, args.model_name_gc if use_gc else args.model_name_gs),
                                     map_location=dev))
    t_f, loss_f, acc_f = run(model, kind)
    print(f"\nAverage time (FIT-GNN - {'coarsened graph' if use_gc else 'subgraph'}): {t_f}\n"
          + (f"Accuracy: {acc_f}" if cls_task else f"L1 loss: {loss_f}"))
    rows = []
    if args.baseline:
        model_b = (network.Classify_graph_gc if cls_task else network.Regress_graph_gc)(args).to(dev)
        model_b.load_state_dict(torch.load(os.path.join(args.path_b, args.model_name_b), map_location=dev))
        t_b, loss_b, acc_b = run(model_b, "orig")
        print(f"Average time (baseline): {t_b}\n" + (f"Accuracy (baseline): {acc_b}" if cls_task else f"L1 loss (baseline): {loss_b}"))
        rows.append(f"{args.dataset},True,{args.experiment},None,None,None,None,None,{args.hidden},{len(ids)},{args.num_layers2},None,{args.lr},{t_b},{loss_b},{acc_b}\n")
    rows.append(f"{args.dataset},False,{args.experiment},{args.exp_setup},{args.coarsening_method},{args.coarsening_ratio},{args.extra_node},"
                f"{args.cluster_node},{args.hidden},{len(ids)},{args.num_layers2},{args.batch_size},{args.lr},{t_f},{loss_f},{acc_f}\n")
    os.makedirs("inference_results", exist_ok=True)
    fn = f"inference_results/{args.task}.csv"
    if not os.path.exists(fn):
        with open(fn, 'w') as f:
            f.write("dataset,baseline,experiment,exp_setup,coarsening_method,coarsening_ratio,extra_node,cluster_node,hidden,"
                    "num_test_samples,num_layers,batch_size,lr,avg_inf_time,avg_loss,acc\n")
    with open(fn, 'a') as f:
        f.writelines(rows)
    return t_f, (acc_f if cls_task else loss_f)


def main(argv=None):
    args = arg_correction(build_parser().parse_args(argv))
    if args.seed is not None:
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    args.train_fitgnn = True
    data, args = train_cli.process_dataset(args)
    if args.task in ("graph_cls", "graph_reg"):
        return graph_inference(args, data)
    from fitgnn_amd import network, pipeline
    from fitgnn_amd.csr import csr_for

    dev = torch.device(args.device)
    rng = np.random.default_rng(args.seed)
    reg = args.task == "node_reg"
    if reg:
        data = pipeline.splits_regression(data, args.train_ratio, args.val_ratio, rng)
    else:
        data = pipeline.splits_classification(data, args.num_classes, args.experiment, rng)
    Model = network.Regress_node if reg else network.Classify_node

    def query_loss(out_row, y):   # node_reg: L1 on the scalar output (inference.py:705-707); node_cls: NLL + hit
        if reg:
            return float((out_row.flatten()[0] - y.float()).abs()), 0
        return float(F.nll_loss(out_row.reshape(1, -1), y.reshape(1))), int(out_row.argmax() == y)
    co = pipeline.coarsening_classification(args, data, 1 - args.coarsening_ratio, args.coarsening_method, device=dev)
    batch = pipeline.build_gs(args, data, co, dev, float_targets=reg)
    ptr = batch.ptr
    # one query = (union row of a cluster's own node, its subgraph); sampled over subgraphs as inference.py:561-634
    core_rows = torch.nonzero(batch.core).flatten().cpu().numpy()
    sub_of_row = np.searchsorted(ptr, core_rows, side="right") - 1
    order = rng.permutation(len(core_rows))[: args.num_test_samples]
    queries = [(int(core_rows[k]), int(sub_of_row[k])) for k in order]
    num = len(queries)

    args.num_layers1 = args.num_layers2  # inference.py builds Net1(..., args.num_layers2, ...)
    model = Model(args).to(dev)
    model.load_state_dict(torch.load(os.path.join(args.path_gs, args.model_name_gs), map_location=dev))
    model.eval()
    ei = batch.edge_index
    cache = {}
    times, losses, hits = [], [], 0
    with torch.no_grad():
        for row, s in queries:
            if s not in cache:  # the subgraph as its own tiny graph: rows ptr[s]:ptr[s+1] of the union
                r0, r1 = int(ptr[s]), int(ptr[s + 1])
                m = (ei[0] >= r0) & (ei[0] < r1)
                cache[s] = (batch.x[r0:r1].contiguous(), (ei[:, m] - r0).contiguous(), r0)
                csr_for(cache[s][1], r1 - r0, "gcn")  # static per-subgraph CSR, built once outside the timed call
            x, e, r0 = cache[s]
            out, dt = timed_forward(model, x, e, dev)
            j = row - r0
            l, h = query_loss(out[j], batch.y[row])
            losses.append(l)
            hits += h
            times.append(dt)
    t_gs = float(np.mean(times[1:])) if len(times) > 1 else float(times[0])
    print(f"\nAverage time (FIT-GNN): {t_gs}\nAccuracy (FIT-GNN): {hits}/{num}")

    rows = []
    if args.baseline:
        model_b = Model(args).to(dev)
        model_b.load_state_dict(torch.load(os.path.join(args.path_b, args.model_name_b), map_location=dev))
        model_b.eval()
        xb = data.x.to(dev).float()
        eb = torch.as_tensor(np.asarray(data.edge_index)).to(dev)
        yb = data.y.flatten().to(dev)
        tb, lb, hb = [], [], 0
        with torch.no_grad():
            for row, _ in queries:
                node = int(batch.node_id[row])
                out, dt = timed_forward(model_b, xb, eb, dev)
                l, h = query_loss(out[node], yb[node])
                lb.append(l)
                hb += h
                tb.append(dt)
        t_b = float(np.mean(tb[1:])) if len(tb) > 1 else float(tb[0])
        print(f"Average time (baseline): {t_b}\nAccuracy (baseline): {hb}/{num}")
        rows.append(f"{args.dataset},True,{args.experiment},None,None,None,None,None,512,{num},{args.num_layers2},None,0.01,{t_b},{np.mean(lb)},{hb / num}\n")
    rows.append(f"{args.dataset},False,{args.experiment},{args.exp_setup},{args.coarsening_method},{args.coarsening_ratio},{args.extra_node},"
                f"{args.cluster_node},512,{num},{args.num_layers2},{args.batch_size},{args.lr},{t_gs},{np.mean(losses)},{hits / num}\n")
    os.makedirs("inference_results", exist_ok=True)
    fn = f"inference_results/{args.task}.csv"
    if not os.path.exists(fn):
        with open(fn, 'w') as f:
            f.write("dataset,baseline,experiment,exp_setup,coarsening_method,coarsening_ratio,extra_node,cluster_node,hidden,"
                    "num_test_samples,num_layers,batch_size,lr,avg_inf_time,avg_loss,acc\n")
    with open(fn, 'a') as f:
        f.writelines(rows)
    return t_gs, hits / num


if __name__ == "__main__":
    main()
