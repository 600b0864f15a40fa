from .evaluation import (all_activity, assign_labels, logreg_fit, logreg_predict, ngram, proportion_weighting,
                         update_ngram_scores)

__all__ = ["assign_labels", "logreg_fit", "logreg_predict", "all_activity", "proportion_weighting", "ngram",
           "update_ngram_scores"]
