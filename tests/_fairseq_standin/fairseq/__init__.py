"""Test double of fairseq's plugin registration contract (see ../README.md). Not fairseq."""
__version__ = "standin"
