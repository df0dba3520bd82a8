"""Callers on the candidate side of the hot path (SURVEY.md section 8 f): feature engineering over the candidate arrays the
covisitation lookup leaves on the device."""
