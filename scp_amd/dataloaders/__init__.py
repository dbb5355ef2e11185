from .ehem_dataset import EHEMDataset  # noqa: F401
