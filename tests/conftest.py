import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from refs import Oracle, oracle_available
    if not oracle_available():
        import __graft_entry__ as g
        g.build()
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """the real reference library, when oracle/_ref was built (it is prebuilt for the GPU box)"""
    from refs import Reference, reference_available
    if not reference_available():
        pytest.skip("oracle/_ref/liblinne_ref.so not built")
    return Reference()


@pytest.fixture(scope="session")
def product():
    """the LINNE public API as exported by liblinne_amd.so"""
    import linne_amd
    from refs import LinneApi
    return LinneApi(linne_amd.LIB_PATH)


@pytest.fixture(scope="session")
def ctx():
    import linne_amd
    c = linne_amd.Context(0, use_torch_stream=False)
    yield c
    c.close()


@pytest.fixture
def ctx_env():
    """context manager: a fresh Context created AND used under the given environment (some knobs are read by
    LINNEAmd_ContextCreate, others per call); the environment is restored afterwards"""
    import contextlib

    @contextlib.contextmanager
    def make(env, scratch_bytes=0):
        import linne_amd
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        c = None
        try:
            c = linne_amd.Context(0, scratch_bytes=scratch_bytes, use_torch_stream=False)
            yield c
        finally:
            if c is not None:
                c.close()
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return make
