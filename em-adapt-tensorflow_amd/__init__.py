"""em-adapt-tensorflow_amd: the EM-Adapt training hot path of xtudbxk/em-adapt-tensorflow on MI355X.

Hand-written HIP (gfx950) kernels behind the reference's own call surface:

* ``estep.estep(...)``                 -- the E-step operator (reference estep.py:18-27)
* ``network.Network`` / ``deeplab.ADAPT`` -- the graph/trainer classes (reference network.py, deeplab.py)
* ``libemadapt_hip.so`` (include/emadapt.h) -- the C ABI, incl. the reference's native ``e_step`` entry

PyTorch is used for device memory, streams and torch.distributed only; all arithmetic on the hot
path runs in the HIP library.  There is no CPU fallback.
"""
__version__ = "0.1.0"
