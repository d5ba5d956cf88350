"""Checkpoint bundles across the two implementations (build container only; imports the REFERENCE in place).

  python tests/golden/make_golden_ckpt.py            writes the fixtures below + runs both directions once

A. reference -> deepof_amd.  For each model (VaDE, VQ-VAE, contrastive with the recurrent encoder; round 4: VaDE and
   contrastive with the TCN family, VaDE and VQ-VAE with the transformer family -- lazily built CensNet tensors and
   BatchNorm buffers in the bundle; deepof_14 graph) the
   reference builds the model and writes a bundle with its OWN ``save_model_info`` (model_utils_new.py:263-329):
   ``tests/golden/ckpt/ref_<model>.pth`` + ``_info.txt`` (data: tensors, a plain rebuild_spec dict, the log summary),
   and ``ref_<model>_io.npz`` = inputs and the reference's eval-mode outputs for them.  tests/test_host_api.py loads the
   bundles with ``deepof_amd.training.load_model_from_ckpt`` and compares the outputs (emulator here, GPU there).
B. deepof_amd -> reference.  ``cross_load_into_reference(path)`` loads a bundle written by deepof_amd's
   ``save_model_info`` with the reference's ``load_model_from_ckpt`` (model_utils_new.py:822-904) and returns the
   reference model's eval outputs; tests/test_host_api.py::test_checkpoint_loads_in_the_reference runs it wherever
   /root/reference exists (this container) and is skipped elsewhere.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
OUT = os.path.join(HERE, "ckpt")


def reference():
    from _ref_import import load_reference
    return load_reference()


def _io(model_name, model, x, a):
    model.eval()
    with torch.no_grad():
        if model_name == "vade":
            dist, z, q, _km = model(x, a)
            return {"z": z.numpy(), "q": q.numpy(), "loc": dist.base_dist.base_dist.loc.numpy()}
        if model_name == "vqvae":
            enc_rec, rec, quant, soft, ze, _ = model(x, a, return_losses=True, return_all_outputs=True)
            return {"ze": ze.numpy(), "soft": soft.numpy(), "quantized": quant.numpy()}
        return {"z": model(x, a).numpy()}


# (model, encoder_type) pairs with a committed reference bundle: ref_<model>.pth (recurrent) / ref_<model>_<enc>.pth
BUNDLES = [("vade", "recurrent"), ("vqvae", "recurrent"), ("contrastive", "recurrent"),
           ("vade", "TCN"), ("contrastive", "TCN"), ("vade", "transformer"), ("vqvae", "transformer")]


def bundle_stem(name, enc):
    return f"ref_{name}" if enc == "recurrent" else f"ref_{name}_{ {'TCN': 'tcn', 'transformer': 'tfm'}[enc] }"


def write_reference_bundles(only_new=False):
    from deepof_amd.graph import adjacency_from_graph, bodypart_graph
    from make_golden import synth_batch
    R = reference()
    os.makedirs(OUT, exist_ok=True)
    nodes, edges = bodypart_graph([""])
    adj = adjacency_from_graph(nodes, edges)
    N, E, T, L, K, B = len(nodes), len(edges), 25, 8, 10, 6
    for name, enc in BUNDLES:
        path = os.path.join(OUT, bundle_stem(name, enc) + ".pth")
        if only_new and os.path.exists(path):
            continue
        torch.manual_seed(401)
        Tm = 2 * T if name == "contrastive" else T
        if name == "vade":
            model = R.M.VaDEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type=enc, kmeans_loss=1.0)
        elif name == "vqvae":
            model = R.M.VQVAEPT((T, N, 3), (T, E, 1), adj, L, K, encoder_type=enc, kmeans_loss=0.0)
        else:
            model = R.M.ContrastivePT((Tm, N, 3), (Tm, E, 1), adj, latent_dim=L, encoder_type=enc)
        x, a = synth_batch(B, T, N, E, 402)
        if enc != "recurrent":
            # the TCN / transformer encoders build their CensNet tensors at the first forward (model_utils_new.py:766-784):
            # a saved model has run, so its bundle holds them; one train-mode pass also moves the BatchNorm buffers off
            # their initial values
            model.eval()
            R.U._materialize_encoder(model, (T, N, 3), (T, E, 1), torch.device("cpu"))
            model.train()
            with torch.no_grad():
                model.encoder(torch.from_numpy(x), torch.from_numpy(a)) if name == "contrastive" else model(torch.from_numpy(x), torch.from_numpy(a))
            model.eval()
        spec = {"model_name": name, "x_shape": (Tm, N, 3), "a_shape": (Tm, E, 1),
                "adjacency_matrix": np.asarray(adj).astype("float32"), "latent_dim": L, "n_components": K,
                "encoder_type": enc, "use_gnn": True, "interaction_regularization": 0.0}
        R.U.save_model_info(path, stage="best_val", epoch=3, train_steps=12, val_total=1.5, model=model,
                            log_summary={"train": {"total_loss": [2.0, 1.5]}, "val": {"total_loss": [2.1, 1.6]}},
                            rebuild_spec=spec, save_weights=True)
        io = _io(name, model, torch.from_numpy(x), torch.from_numpy(a))
        np.savez_compressed(os.path.join(OUT, bundle_stem(name, enc) + "_io.npz"), x=x, a=a, **io)
        print(name, enc, os.path.getsize(path), "bytes;", sorted(io))


def cross_load_into_reference(path, x, a):
    """Load ``path`` (a deepof_amd bundle) with the reference's load_model_from_ckpt; -> (outputs, load_report)."""
    R = reference()
    model, _logs, spec, report = R.U.load_model_from_ckpt(path, device=torch.device("cpu"), strict=False)
    return _io(str(spec["model_name"]).lower(), model, torch.from_numpy(x), torch.from_numpy(a)), report


if __name__ == "__main__":
    write_reference_bundles(only_new="new" in sys.argv[1:])
