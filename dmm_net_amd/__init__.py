def use_shipped_miopen_db(develop: bool = False, enable: bool = True):
    """Seed MIOpen's USER find-db / perf-db with the solver choices shipped in ``dmm_net_amd/miopen_db`` (what MIOpen's own
    search picked on an MI355X for the encoder shapes of BASELINE configs 3 and 4 and the frame loop; plain-text files
    keyed by problem, entries for other MIOpen builds are simply ignored).

    MIOpen WRITES to its user db (every new shape it searches), and the variable is process wide -- so the shipped files
    are never handed to it directly: they are copied once into a per-user cache directory
    (``~/.cache/dmm_net_amd/miopen_db-<fingerprint>``, several ranks may race: files are put in place atomically) and
    MIOPEN_USER_DB_PATH points there.  MIOpen's OWN variable, MIOPEN_USER_DB_PATH, always wins when the caller has set it;
    this package reads no variable of its own: ``use_shipped_miopen_db(enable=False)`` (before the first convolution) takes
    the shipped db out again, ``develop=True`` points MIOpen at the tracked directory itself -- only to refresh what is
    shipped (tools/).  Must run before the first convolution: the encoders (``FeatureEncoder``, ``FastEncoder``) call it from
    their constructors; importing the package does not."""
    import os
    import shutil
    global _DB_SET_BY_US
    if not enable:
        if _DB_SET_BY_US and os.environ.get("MIOPEN_USER_DB_PATH") == _DB_SET_BY_US:
            del os.environ["MIOPEN_USER_DB_PATH"]
        _DB_SET_BY_US = None
        return None
    if os.environ.get("MIOPEN_USER_DB_PATH") and os.environ.get("MIOPEN_USER_DB_PATH") != _DB_SET_BY_US:
        return os.environ.get("MIOPEN_USER_DB_PATH")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
    if not os.path.isdir(src):
        return None
    if develop:
        os.environ["MIOPEN_USER_DB_PATH"] = _DB_SET_BY_US = src
        return src
    files = sorted(f for f in os.listdir(src) if os.path.isfile(os.path.join(src, f)))
    import hashlib
    h = hashlib.sha256()                     # names + CONTENT: an update of the same size must not keep a stale copy
    for f in files:
        h.update(f.encode() + b"\0")
        with open(os.path.join(src, f), "rb") as fh:
            h.update(fh.read())
    tag = h.hexdigest()[:12]
    dst = os.path.join(os.path.expanduser("~"), ".cache", "dmm_net_amd", "miopen_db-" + tag)
    try:
        os.makedirs(dst, exist_ok=True)
        for f in files:
            if not os.path.exists(os.path.join(dst, f)):
                tmp = os.path.join(dst, f".{f}.{os.getpid()}.tmp")
                shutil.copy(os.path.join(src, f), tmp)
                os.replace(tmp, os.path.join(dst, f))
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = _DB_SET_BY_US = dst
    return dst


_DB_SET_BY_US = None
# (NOT called at import: importing the package -- e.g. only ``dmm_net_amd.match_model`` for the one-line swap -- leaves the
# environment and ~/.cache alone.  The encoders call it when they are constructed: ``FeatureEncoder.__init__`` /
# ``FastEncoder.__init__``, before the process's first convolution of theirs.)
