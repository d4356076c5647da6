"""TEST INFRASTRUCTURE: stand-in for OpenCV (opencv-python==4.8.1.78 in the reference's requirements.txt), which the image lacks.
Nothing here resamples: the one call that answers is `resize` to the source's OWN size, where OpenCV returns the source values
unchanged (scale factors 1: every output pixel centre falls on an input pixel centre) -- enough to run the reference's
`_simulate_haze` on a cirrus map that already has the patch size.  Code paths that need real resampling (the motion-blur
kernel rotation: getRotationMatrix2D + warpAffine) raise and stay unpinned by reference-generated fixtures."""
INTER_LINEAR = 1


def resize(src, dsize, interpolation=INTER_LINEAR):
    w, h = dsize
    if tuple(src.shape[:2]) != (h, w):
        raise NotImplementedError("cv2 stand-in: resize only to the source's own size")
    return src.copy()


def getRotationMatrix2D(*a, **k):
    raise NotImplementedError("cv2 stand-in: no rotation / warping")


def warpAffine(*a, **k):
    raise NotImplementedError("cv2 stand-in: no rotation / warping")
