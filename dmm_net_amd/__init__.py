

def _point_miopen_at_the_repo_db():
    """MIOpen's user find-db / perf-db for this package: dmm_net_amd/miopen_db holds the solver choices MIOpen's own
    search made on an MI355X for the encoder shapes of BASELINE configs 3 and 4 (plain-text files keyed by problem;
    entries for other MIOpen builds are simply ignored).  Must be in the environment before MIOpen initialises, i.e.
    before the first convolution; an explicit MIOPEN_USER_DB_PATH wins.  A read-only install gets a private copy."""
    import os
    import shutil
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
    if not os.path.isdir(src):
        return
    dst = src
    if not os.access(src, os.W_OK):
        dst = os.path.join(os.path.expanduser("~"), ".cache", "dmm_net_amd", "miopen_db")
        try:
            os.makedirs(dst, exist_ok=True)
            for f in os.listdir(src):
                if not os.path.exists(os.path.join(dst, f)):
                    shutil.copy(os.path.join(src, f), dst)
        except OSError:
            return
    os.environ["MIOPEN_USER_DB_PATH"] = dst


_point_miopen_at_the_repo_db()
