"""dmae_vtp (MI355X path): DMAE retrieval pieces of SURVEY.md section 8a (T11b, L5)."""
