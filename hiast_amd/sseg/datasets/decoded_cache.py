"""Decoded-image cache of the training data path (no counterpart in the reference, off by default).

A self-training round reads every target image ~20 times (8000 iterations x batch 8 over 2975 Cityscapes images, twice
per sample with CopyPaste), and the PNG decode of a 2048x1024 image + its pseudo-label is half of a worker's time per
sample (measured: 82 of 164 ms).  With `cfg.dataset.decoded_cache_dir` set (e.g. a directory on /dev/shm), load_data
stores each decoded uint8 array once as a raw .npy file and later reads are a memory copy.  Files are keyed by path +
size + mtime of the source (a rewritten pseudo-label is a new entry), written to a temporary name and renamed
(workers race benignly), and the directory is kept below `cfg.dataset.decoded_cache_gb` (each process counts what it
sees: the bound is approximate to within the files written concurrently)."""
import hashlib
import os

import numpy as np


class DecodedCache:

    def __init__(self, directory, max_gb=16.0):
        self.dir = directory
        self.max_bytes = int(float(max_gb) * (1 << 30))
        os.makedirs(directory, exist_ok=True)
        self._seen = None           # bytes in the directory as of the last scan + own writes since
        self._puts = 0

    def _key(self, path, salt=""):
        st = os.stat(path)
        h = hashlib.sha1(("%s|%d|%d|%s" % (os.path.abspath(path), st.st_size, st.st_mtime_ns, salt)).encode()).hexdigest()
        return os.path.join(self.dir, h + ".npy")

    def _usage(self):
        if self._seen is None or self._puts % 64 == 0:
            total = 0
            with os.scandir(self.dir) as it:
                for e in it:
                    try:
                        total += e.stat().st_size
                    except OSError:
                        pass
            self._seen = total
        return self._seen

    def get(self, path, salt=""):
        """-> writable uint8 array (a private copy), or None"""
        f = self._key(path, salt)
        try:
            return np.array(np.load(f, mmap_mode="r"))
        except (OSError, ValueError):       # absent, or a torn file from a crashed writer
            return None

    def put(self, path, arr, salt=""):
        if self._usage() + arr.nbytes > self.max_bytes:
            return False
        f = self._key(path, salt)
        tmp = "%s.%d.tmp" % (f, os.getpid())
        try:
            with open(tmp, "wb") as fh:
                np.save(fh, np.ascontiguousarray(arr))
            os.replace(tmp, f)
        except OSError:                     # full / read-only file system: the cache is best effort
            try:
                os.remove(tmp)
            except OSError:
                pass
            return False
        self._seen += arr.nbytes
        self._puts += 1
        return True

    def load(self, path, decode, salt=""):
        """array for `path`: from the cache, else `decode(path)` (stored for next time).  `salt` separates different
        decodings of one file (e.g. the label id map of a dataset class)"""
        arr = self.get(path, salt)
        if arr is None:
            arr = decode(path)
            if isinstance(arr, np.ndarray):
                self.put(path, arr, salt)
        return arr
