"""Replacements for the reference's dense-adjacency modules (models/pytorch/)."""
