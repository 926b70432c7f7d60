from .global_retrieval_recall import GlobalRetrievalRecall  # noqa: F401
