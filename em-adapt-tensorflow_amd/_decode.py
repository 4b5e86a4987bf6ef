"""Decode worker of the input pipeline (data.dataset): a separate PROCESS that turns (JPEG path, PNG label path) into uint8 pixels
inside a shared-memory staging ring.  Imports only the standard library, NumPy and PIL -- never torch or the HIP library -- so a
worker starts in a fraction of a second and nothing in it can touch the GPU.

Why processes: PIL spends ~0.45 ms of every sample inside the interpreter (header parsing, PNG chunk walking, array conversion);
decode threads therefore serialise on the interpreter lock at ~2,200 samples/s however many there are, and every trip the
training step's launch code makes through that lock queues behind them (profiles/r06_loader.txt).  A process has its own lock.

Protocol (line-based, one request / one reply, over the worker's stdin / stdout):
    request   "<img_off>\t<lab_off>\t<cap>\t<img_path>\t<label_path>\n"     offsets into the shared block, cap = bytes available
    reply     "OK <h> <w>\n"  |  "ERR <message>\n"
    the image lands at img_off as uint8 [h][w][3] RGB, the label at lab_off as uint8 [h][w] (palette index = class id,
    dataset.py:139-141 decode_png of the SegmentationClassAug files).
Started as  python -m em_adapt_tensorflow_amd._decode <shared-memory name>  and ends when stdin closes.
"""
import sys


def decode_into(buf, img_off, lab_off, cap, img_f, label_f):
    """Decode one sample into buf (a writable uint8 NumPy view of the staging block).  Returns (h, w)."""
    import numpy as np
    from PIL import Image
    with Image.open(img_f) as im:
        img = np.asarray(im.convert("RGB"), dtype=np.uint8)
    with Image.open(label_f) as lm:
        lab = np.asarray(lm if lm.mode in ("L", "P") else lm.convert("L"), dtype=np.uint8)
    h, w = img.shape[:2]
    if lab.shape[:2] != (h, w):
        raise ValueError("label %s is %s, image %s is %s" % (label_f, lab.shape[:2], img_f, (h, w)))
    if h * w * 3 > lab_off - img_off or h * w > cap - (lab_off - img_off):
        raise ValueError("%s is %d x %d: larger than the staging slot (config max_sample_pixels)" % (img_f, h, w))
    buf[img_off:img_off + h * w * 3] = img.reshape(-1)
    buf[lab_off:lab_off + h * w] = lab.reshape(-1)
    return h, w


def main(argv):
    import numpy as np
    from multiprocessing import shared_memory
    # (the parent owns the block's lifetime; keep this process's resource tracker out of it)
    try:
        shm = shared_memory.SharedMemory(name=argv[1], track=False)
    except TypeError:   # Python < 3.13: no `track`; unregister by hand
        shm = shared_memory.SharedMemory(name=argv[1])
        try:
            from multiprocessing import resource_tracker
            resource_tracker.unregister(shm._name, "shared_memory")
        except Exception:
            pass
    buf = np.ndarray((shm.size,), dtype=np.uint8, buffer=shm.buf)
    out = sys.stdout
    try:
        for line in sys.stdin:
            parts = line.rstrip("\n").split("\t")
            try:
                h, w = decode_into(buf, int(parts[0]), int(parts[1]), int(parts[2]), parts[3], parts[4])
                out.write("OK %d %d\n" % (h, w))
            except Exception as e:   # the parent raises it with the sample's name
                out.write("ERR %s: %s\n" % (type(e).__name__, str(e).replace("\n", " ")))
            out.flush()
    finally:
        del buf
        shm.close()


if __name__ == "__main__":
    main(sys.argv)
