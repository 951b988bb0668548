"""Test infrastructure (never imported by the product): a RECORDING stand-in for the `cv2` module, used in the build container to pin what the
reference's data pipeline hands to OpenCV (cv2 itself is absent from the image, so no pixel vector of the reference's decode / resize / warp
can exist - but every ARGUMENT can be recorded: datasets/author_hw_dataset.py:364-404 `imread` + `resize(..., fx, fy, INTER_CUBIC)`,
utils/augmentation.py:61-72 `warpAffine(img, matrix, shape, borderValue=255)`).

The stand-in returns arrays of the size OpenCV documents for the call, so the reference's control flow (which depends on shapes only) runs
unchanged; the pixel values it returns are NOT OpenCV's. Sizes: resize with dsize (0, 0) -> Size(round(fx * cols), round(fy * rows)), cvRound =
round-half-to-even; warpAffine -> dsize. Constants carry OpenCV's published enum values."""
import numpy as np

CONSTANTS = {"INTER_NEAREST": 0, "INTER_LINEAR": 1, "INTER_CUBIC": 2, "INTER_AREA": 3, "INTER_LANCZOS4": 4, "BORDER_CONSTANT": 0, "BORDER_REPLICATE": 1,
             "WARP_INVERSE_MAP": 16, "IMREAD_GRAYSCALE": 0, "THRESH_BINARY": 0, "THRESH_OTSU": 8, "MORPH_ELLIPSE": 2, "COLOR_BGR2HSV": 40, "COLOR_HSV2BGR": 54}
INTER_NAMES = {0: "INTER_NEAREST", 1: "INTER_LINEAR", 2: "INTER_CUBIC", 3: "INTER_AREA", 4: "INTER_LANCZOS4"}


class RecordingCv2:
    def __init__(self, root):
        self.root = root            # recorded paths are relative to it
        self.calls = []
        for k, v in CONSTANTS.items():
            setattr(self, k, v)

    def imread(self, path, flags=1):
        import os
        from PIL import Image
        self.calls.append(["imread", os.path.relpath(path, self.root), int(flags)])
        if not os.path.exists(path):
            return None
        return np.asarray(Image.open(path).convert("L" if flags == 0 else "RGB"))

    def resize(self, src, dsize, dst=None, fx=0.0, fy=0.0, interpolation=1):
        from PIL import Image
        if tuple(dsize) == (0, 0):
            w, h = int(round(fx * src.shape[1])), int(round(fy * src.shape[0]))
        else:
            w, h = dsize
        self.calls.append(["resize", list(src.shape[:2]), list(dsize), float(fx), float(fy), INTER_NAMES[int(interpolation)], [h, w]])
        return np.asarray(Image.fromarray(src).resize((w, h), Image.BICUBIC))

    def warpAffine(self, src, M, dsize, dst=None, flags=1, borderMode=0, borderValue=0):
        M = np.asarray(M, dtype=np.float64)
        self.calls.append(["warpAffine", list(src.shape[:2]), [float(v) for v in M.reshape(-1)], [int(dsize[0]), int(dsize[1])],
                           INTER_NAMES[int(flags) & 7], bool(int(flags) & CONSTANTS["WARP_INVERSE_MAP"]), int(borderMode), float(np.ravel(borderValue)[0])])
        return np.full((int(dsize[1]), int(dsize[0])) + tuple(src.shape[2:]), float(np.ravel(borderValue)[0]), dtype=src.dtype)

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        raise NotImplementedError("cv2.%s is not on the path of the shipped configs' data pipeline" % k)
