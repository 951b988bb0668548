"""Import the (read-only, unmodified) reference from /root/reference inside THIS container.
Only used by tools/gen_golden.py to produce tests/golden/*; it never travels to the GPU box."""
import importlib.machinery as M
import os
import sys
import types

REF = "/root/reference"


def bootstrap():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)  # configs use relative data/ paths

    class Stub(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return 0

    for n in ["cv2", "skimage", "skimage.draw", "skimage.morphology", "skimage.filters", "torchvision", "torchvision.utils", "editdistance"]:
        m = Stub(n)
        m.__spec__ = M.ModuleSpec(n, None)
        m.__path__ = []
        sys.modules[n] = m
    ds = types.ModuleType("datasets")
    ds.__path__ = [REF + "/datasets"]
    sys.modules["datasets"] = ds
