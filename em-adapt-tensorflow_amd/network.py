"""``Network`` base class with the reference's surface (network.py:7-110), backed by the HIP library.

The reference builds a TF-1.x graph and runs it through ``tf.Session``; here ``build()`` lays out a
static execution plan (layer list, flat parameter / gradient / momentum buffers in HBM, activation
buffers per batch size) and the "session runs" of the training loop become methods:

    sess.run(net["accum_gradient_accum"], feed)  ->  net["accum_gradient_accum"](images, labels, keep_prob)
    sess.run(net["accum_gradient_update"]) ; sess.run(net["accum_gradient_clean"])  ->  same names, callables

State dicts keep the reference's names: ``net``, ``weights``, ``loss``, ``saver``, ``trainable_list``.
"""
import os

import numpy as np
import torch

from . import ops


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
        for name in ("norm", "lr", "best"):
            self.saver[name] = _Saver(self)

    def restore_from_model(self, saver, model_path, checkpoint=False):
        """network.py:104-110."""
        if checkpoint:
            cands = sorted(p for p in os.listdir(model_path) if p.endswith(".npz"))
            if not cands:
                raise FileNotFoundError("no checkpoint in %s" % model_path)
            model_path = os.path.join(model_path, cands[-1])
        saver.restore(None, model_path)


class _Saver:
    """Stand-in for tf.train.Saver over trainable_list (network.py:100-102).  Unlike the reference it also
    stores the momentum slots, so a resumed run continues exactly (checkpoint row)."""

    def __init__(self, net):
        self.netobj = net

    def save(self, sess, path, global_step=None):
        n = self.netobj
        if global_step is not None:
            path = "%s-%d" % (path, int(global_step))
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        blob = {}
        for layer, (w, b) in n.weights.items():
            blob[layer + "_weights"] = w.detach().cpu().numpy()
            blob[layer + "_bias"] = b.detach().cpu().numpy()
        if getattr(n, "_flat", None) is not None:
            blob["__momentum__"] = n._flat["momentum"].detach().cpu().numpy()
        np.savez(path + ".npz", **blob)
        return path + ".npz"

    def restore(self, sess, path):
        n = self.netobj
        if not path.endswith(".npz"):
            path = path + ".npz"
        with np.load(path) as z:
            for layer, (w, b) in n.weights.items():
                w.copy_(torch.from_numpy(z[layer + "_weights"]).to(w.device))
                b.copy_(torch.from_numpy(z[layer + "_bias"]).to(b.device))
            if "__momentum__" in z.files and getattr(n, "_flat", None) is not None:
                n._flat["momentum"].copy_(torch.from_numpy(z["__momentum__"]).to(n._flat["momentum"].device))


__all__ = ["Network", "ops", "round_up"]
