"""MI355X-native hot path of LingmaFuture/PRCV2025REID."""
__version__ = "0.1.0"
