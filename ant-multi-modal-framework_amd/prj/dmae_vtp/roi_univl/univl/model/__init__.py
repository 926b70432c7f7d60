import os

__path__.append(os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", "..", "..", "base_vtp", "roi_univl", "univl", "model")))
