"""TEST INFRASTRUCTURE: stand-in for OpenCV, which the image lacks.  Only importability is provided: nothing here computes,
so reference code paths that need cv2 (motion-blur kernel rotation, haze-map resize) are NOT pinned by fixtures."""
INTER_LINEAR = 1
