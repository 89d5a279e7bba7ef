"""Source-level guards for two classes of defect the GPU fuzzers have found or would not find reliably:
* device memory the library allocates must come from rlr::dev_malloc, so that RLR_POISON_ALLOC=1 covers every buffer;
* a null-stream fill / device-to-device copy of device memory may return before it has run and is NOT ordered
  against the non-blocking streams searches run on (the histogram defect of round 1), so each one must be followed
  by an explicit wait before the function goes on, or be issued Async on the consumer's own stream."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = sorted(glob.glob(os.path.join(ROOT, "rust-local-rag_amd", "csrc", "*.hip")) +
                 glob.glob(os.path.join(ROOT, "rust-local-rag_amd", "csrc", "*.cpp")))


def _lines(path):
    with open(path) as f:
        return f.read().split("\n")


def test_every_device_allocation_goes_through_dev_malloc():
    assert SOURCES
    bare = []
    for path in SOURCES:
        for i, line in enumerate(_lines(path), 1):
            code = line.split("//")[0]
            if re.search(r"(?<![A-Za-z_])hipMalloc\(", code):
                bare.append((os.path.basename(path), i))
    # the one call inside rlr::dev_malloc itself
    assert len(bare) == 1 and bare[0][0] == "index.hip", bare


def test_null_stream_fills_and_device_copies_are_followed_by_a_wait():
    waits = ("hipStreamSynchronize(nullptr)", "hipDeviceSynchronize()", "hipMemcpyDeviceToHost", "hipMemcpyHostToDevice")
    loose = []
    for path in SOURCES:
        lines = _lines(path)
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            fill = re.search(r"(?<![A-Za-z_])hipMemset\(", code)
            d2d = re.search(r"(?<![A-Za-z_])hipMemcpy\(", code) and "hipMemcpyDeviceToDevice" in " ".join(lines[i:i + 3])
            if not (fill or d2d):
                continue
            # a wait (or a synchronous host copy on the same null stream) within the next few statements
            window = " ".join(l.split("//")[0] for l in lines[i + 1:i + 9])
            if not any(w in window for w in waits):
                loose.append((os.path.basename(path), i + 1, code.strip()))
    assert not loose, loose
