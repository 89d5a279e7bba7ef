"""rust-local-rag_amd -- MI355X-native search_documents hot path for rust-local-rag.

The directory name is not a Python identifier; import it with
    importlib.import_module("rust-local-rag_amd")
(tests/conftest.py, bench.py and __graft_entry__.py do exactly that).

Contents: csrc/ (HIP kernels, C ABI, host engine -> librlr_gpu.so), the ctypes view of
that library and the host-side mirror of the reference's RagEngine search interface.
Importing the package loads the library and raises if it has not been built: there is no
CPU implementation of the search path.
"""
from ._native import (DEFAULT_DIVERSITY, DEFAULT_TOP_K, MAX_TOP_K, RLR_F16, RLR_F32, RlrError, SO_PATH, lib)
from .engine import (DocumentChunk, QueryWeights, RagEngine, ResolvedWeights, SearchRequest, SearchResult,
                     format_search_results, normalize, resolve_weight)
from .index import GpuIndex, MultiGpuIndex, Profile, default_guard_eps, device_count
from .lexical import LexicalIndex, tokenize
from .persistence import (LoadReport, get_index_path, get_legacy_path, get_sidecar_path, load_from_disk,
                          sanitize_model_name, save_sidecar, save_to_disk)

lib()  # fail loudly at import time when librlr_gpu.so is missing

__all__ = [
    "DEFAULT_DIVERSITY", "DEFAULT_TOP_K", "MAX_TOP_K", "RLR_F16", "RLR_F32", "RlrError", "SO_PATH", "lib",
    "DocumentChunk", "QueryWeights", "RagEngine", "ResolvedWeights", "SearchRequest", "SearchResult",
    "format_search_results", "normalize", "resolve_weight", "GpuIndex", "MultiGpuIndex", "Profile", "default_guard_eps",
    "device_count", "LexicalIndex", "tokenize", "LoadReport", "get_index_path", "get_legacy_path", "get_sidecar_path",
    "load_from_disk", "sanitize_model_name", "save_sidecar", "save_to_disk",
]
