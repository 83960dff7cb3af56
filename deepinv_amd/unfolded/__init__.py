from .unfolded import BaseUnfold, unfolded_builder
