"""MI355X-native IRFinder read-block counting path."""
__version__ = "0.1.0"
