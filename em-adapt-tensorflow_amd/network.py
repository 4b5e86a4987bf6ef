"""``Network`` base class with the reference's surface (network.py:7-110), backed by the HIP library.

The reference builds a TF-1.x graph and runs it through ``tf.Session``; here ``build()`` lays out a
static execution plan (layer list, flat parameter / gradient / momentum buffers in HBM, activation
buffers per batch size) and the "session runs" of the training loop become methods:

    sess.run(net["accum_gradient_accum"], feed)  ->  net["accum_gradient_accum"](images, labels, keep_prob)
    sess.run(net["accum_gradient_update"]) ; sess.run(net["accum_gradient_clean"])  ->  same names, callables

State dicts keep the reference's names: ``net``, ``weights``, ``loss``, ``saver``, ``trainable_list``.
"""
import json
import os
import re
import warnings

import numpy as np
import torch

from . import ops
from .data import dist_rank_world


def round_up(v, m):
    return (v + m - 1) // m * m


class Network:
    def __init__(self, config):
        # network.py:8-20
        self.config = config
        self.h, self.w = self.config.get("input_size", (25, 25))
        self.category_num = self.config.get("category_num", 21)
        self.accum_num = self.config.get("accum_num", 1)
        self.data = self.config.get("data", None)
        self.net = {}
        self.weights = {}
        self.trainable_list = []
        self.loss = {}
        self.images = {}
        self.saver = {}
        dev = self.config.get("device", None)
        if dev is None:
            if not torch.cuda.is_available():
                raise RuntimeError("em-adapt-tensorflow_amd needs an MI355X GPU: there is no CPU execution path")
            dev = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(dev)

    # -- graph construction (network.py:22-48) ------------------------------------------------
    def build(self):
        if "output" not in self.net:
            self.net["input"] = None   # placeholders: filled by feed (network.py:24-26)
            self.net["label"] = None
            self.net["output"] = self.create_network()
            self.pred()
        return self.net["output"]

    def create_network(self, layer=None):  # "need to rewrite" in the reference (network.py:33)
        if "init_model_path" in self.config:
            self.load_init_model()
        return layer

    def pred(self):
        """network.py:39-41: net["rescale_output"] = resize_bilinear(output,(h,w)); net["pred"] = argmax(axis=3).
        Lazily evaluated by ``predict_labels``; the training hot path never runs it."""
        self.net["rescale_output"] = None
        self.net["pred"] = None

    def predict_labels(self, logits=None, want_rescale=False):
        """Evaluate the inference head on the current (or given) fc8 logits [B,s,s,ldl]: int64 [B,h,w]."""
        if logits is None or logits is self.net.get("output"):
            logits = self._bufs["act"][self._plan[-1].name]  # the class-padded fc8 buffer behind net["output"]
        elif not logits.is_contiguous():
            logits = logits.contiguous()
        pred, resc = ops.pred_bilinear_argmax(logits, self.category_num, self.h, self.w, want_rescale)
        self.net["pred"] = pred
        self.net["rescale_output"] = resc
        return pred

    def load_init_model(self):
        pass

    def get_weights_and_biases(self, layer):
        if layer in self.weights:
            return self.weights[layer]
        raise KeyError(layer)

    def predict(self):
        raise NotImplementedError("Network.predict is dead code in the reference (network.py:65-67 use undefined names)")

    # -- training setup (network.py:91-102) ---------------------------------------------------
    def pre_train(self, base_lr, weight_decay, momentum, batch_size, save_layers=("input", "output", "label", "pred")):
        # accumulators = zeros_like(trainable_list) (network.py:92) live in the flat `accum` buffer
        self.getloss(weight_decay)
        self.optimize(base_lr, momentum)
        # network.py:100-102: Saver(max_to_keep=2) for "norm" and "best", the TF default (5) for "lr"
        for name, keep in (("norm", 2), ("lr", 5), ("best", 2)):
            self.saver[name] = _Saver(self, max_to_keep=keep)

    def restore_from_model(self, saver, model_path, checkpoint=False):
        """network.py:104-110.  checkpoint=True: `model_path` is a directory and the checkpoint to load is the one its
        ``checkpoint`` index file names (tf.train.get_checkpoint_state: the LAST one written there)."""
        if checkpoint:
            model_path = latest_checkpoint(model_path)
        saver.restore(None, model_path)


_INDEX = "checkpoint"   # the index file tf.train.Saver keeps next to its checkpoints
_LEGACY_MOMENTUM = "__momentum__"   # round-1 checkpoints: the flat momentum buffer as a single array


def _step_of(path):
    m = re.search(r"-(\d+)\.npz$", path)
    return int(m.group(1)) if m else -1


def latest_checkpoint(model_dir):
    """tf.train.get_checkpoint_state(dir).model_checkpoint_path: the checkpoint written last into `model_dir`, from
    the ``checkpoint`` index file; directories without one (hand-copied files) fall back to the highest global step,
    compared as a NUMBER ("norm-11999" is newer than "norm-5999")."""
    index = os.path.join(model_dir, _INDEX)
    if os.path.exists(index):
        with open(index) as f:
            state = json.load(f)
        path = state["model_checkpoint_path"]
        path = path if os.path.isabs(path) else os.path.join(model_dir, path)
        if os.path.exists(path):
            return path
    cands = [p for p in os.listdir(model_dir) if p.endswith(".npz")]
    if not cands:
        raise FileNotFoundError("no checkpoint in %s" % model_dir)
    cands.sort(key=lambda p: (_step_of(p), os.path.getmtime(os.path.join(model_dir, p))))
    return os.path.join(model_dir, cands[-1])


class _Saver:
    """Stand-in for tf.train.Saver over trainable_list (network.py:100-102): ``save(sess, path, global_step)`` writes
    ``<path>-<step>.npz`` and keeps the last `max_to_keep` files THIS saver wrote (TF semantics: "norm" and "best"
    keep 2, "lr" the default 5).  Writes are atomic (temp file + rename) and recorded in a ``checkpoint`` index file
    like TF's, which ``restore_from_model(checkpoint=True)`` follows.  Unlike the reference it also stores the
    momentum slots, per layer in the layer's own (unpadded) shape like the weights, so a resumed run continues exactly
    and a checkpoint does not depend on the buffer padding of the run that wrote it."""

    def __init__(self, net, max_to_keep=5):
        self.netobj = net
        self.max_to_keep = max_to_keep
        self._kept = []

    def _momentum_views(self, layer):
        n = self.netobj
        seg = n._segments[layer]
        s, nw, nb = seg["start"], seg["nw"], seg["nb"]
        kh, kw, cin, cout = seg["shape"]
        v = n._flat["momentum"]
        return v[s:s + nw].view(seg["pshape"])[:, :, :cin, :cout], v[s + nw:s + nw + nb][:cout]

    def save(self, sess, path, global_step=None):
        n = self.netobj
        if global_step is not None:
            path = "%s-%d" % (path, int(global_step))
        final = os.path.abspath(path + ".npz")
        if dist_rank_world(n.config)[0] != 0:   # data-parallel runs: the replicas are identical, rank 0 writes
            return final
        os.makedirs(os.path.dirname(final), exist_ok=True)
        blob = {}
        for layer, (w, b) in n.weights.items():
            blob[layer + "_weights"] = w.detach().cpu().numpy()
            blob[layer + "_bias"] = b.detach().cpu().numpy()
            if getattr(n, "_flat", None) is not None:
                vw, vb = self._momentum_views(layer)
                blob[layer + "_weights/Momentum"] = vw.detach().cpu().numpy()
                blob[layer + "_bias/Momentum"] = vb.detach().cpu().numpy()
        tmp = "%s.tmp.%d" % (final, os.getpid())
        with open(tmp, "wb") as f:
            np.savez(f, **blob)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, final)
        if final in self._kept:
            self._kept.remove(final)
        self._kept.append(final)
        while self.max_to_keep and len(self._kept) > self.max_to_keep:
            old = self._kept.pop(0)
            if os.path.exists(old):
                os.remove(old)
        d = os.path.dirname(final)
        index_tmp = os.path.join(d, "%s.tmp.%d" % (_INDEX, os.getpid()))
        with open(index_tmp, "w") as f:
            json.dump({"model_checkpoint_path": os.path.basename(final),
                       "all_model_checkpoint_paths": [os.path.basename(p) for p in self._kept
                                                      if os.path.dirname(p) == d]}, f)
        os.replace(index_tmp, os.path.join(d, _INDEX))
        return final

    def restore(self, sess, path):
        n = self.netobj
        if not path.endswith(".npz"):
            path = path + ".npz"
        flat = getattr(n, "_flat", None)
        with np.load(path) as z:
            slots = 0
            for layer, (w, b) in n.weights.items():
                w.copy_(torch.from_numpy(z[layer + "_weights"]).to(w.device))
                b.copy_(torch.from_numpy(z[layer + "_bias"]).to(b.device))
                if flat is not None and layer + "_weights/Momentum" in z.files:
                    vw, vb = self._momentum_views(layer)
                    vw.copy_(torch.from_numpy(z[layer + "_weights/Momentum"]).to(vw.device))
                    vb.copy_(torch.from_numpy(z[layer + "_bias/Momentum"]).to(vb.device))
                    slots += 1
            if flat is not None and slots == 0:
                # round-1 files hold the whole (padded) momentum buffer as one array: usable when the layout still matches
                if _LEGACY_MOMENTUM in z.files and z[_LEGACY_MOMENTUM].size == flat["momentum"].numel():
                    flat["momentum"].copy_(torch.from_numpy(z[_LEGACY_MOMENTUM]).reshape(-1).to(flat["momentum"].device))
                else:
                    why = ("its %r array does not fit this run's buffer layout" % _LEGACY_MOMENTUM
                           if _LEGACY_MOMENTUM in z.files else "it holds no momentum slots")
                    warnings.warn("checkpoint %s: %s -- weights restored, momentum left as it is (zero in a fresh run), "
                                  "as with the reference's weights-only Saver" % (path, why))
            elif flat is not None and slots != len(n.weights):
                raise ValueError("checkpoint %s holds momentum slots for %d of %d layers" % (path, slots, len(n.weights)))
        if flat is not None:
            n._flat["params"].add_(0)   # bump torch's version counter: derived filter images are refreshed


__all__ = ["Network", "ops", "round_up", "latest_checkpoint"]
