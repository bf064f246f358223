"""mlvectordb_amd -- MI355X-native exhaustive kNN behind MLVectorDB's Index / QueryProcessor surface.

Only the search hot path is here: ``Index`` (drop-in for the reference's
hnswlib-backed index), ``QueryProcessor.find_similar`` dispatch, the row/query carriers, and
the ctypes binding onto the HIP library in ``csrc/``.  Importing the package does not load the
library; constructing an ``Index`` row store does, and fails loudly if it is not built or no
GPU is present.
"""
from .interfaces import (IndexProtocol, QueryProcessorProtocol, SearchResultProtocol, VectorDTO,  # noqa: F401
                         VectorProtocol)
from .vector import Vector  # noqa: F401
from .simple_vector import SimpleVector  # noqa: F401
from .index import BatchHits, Index, SearchResult  # noqa: F401
from .query_processor import QueryProcessor  # noqa: F401
from .storage import ArrayStorage, InMemoryStorage  # noqa: F401
from .engine import HipScanEngine, ScanEngine  # noqa: F401

__all__ = ["Index", "SearchResult", "BatchHits", "QueryProcessor", "InMemoryStorage", "ArrayStorage", "Vector", "SimpleVector", "VectorDTO",
           "VectorProtocol", "IndexProtocol", "SearchResultProtocol", "QueryProcessorProtocol",
           "HipScanEngine", "ScanEngine"]
