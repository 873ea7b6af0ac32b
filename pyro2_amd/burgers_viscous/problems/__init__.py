__all__ = ["converge", "test", "tophat"]
