"""Drop-in alias: `import star.models.multiagent_mae`, `star.utils.CoModule`, ... resolve to the MI355X
implementation in `star_amd` under the reference's own module paths (completion/*.py import these names)."""
