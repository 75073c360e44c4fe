"""Stand-in for OpenCV (oracle harness only): the reference's visualiser
imports cv2 at module import time; rendering is never exercised."""


def destroyAllWindows():
    pass
