"""peneo_amd — MI355X-native forward/backward hot path of PEneo."""
__version__ = "0.1.0"
