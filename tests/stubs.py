"""the boundary callees' stand-ins under the names the tests' configs use (``tests.stubs.StubTextEncoder`` ...): they live in
``adaprompt_amd.standins`` so that bench.py does not import from the test tree."""
from adaprompt_amd.standins import StubEmbeddingManager, StubTextEncoder          # noqa: F401
