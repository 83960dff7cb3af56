from .unfolded import BaseUnfold, unfolded_builder, DEQ_builder
