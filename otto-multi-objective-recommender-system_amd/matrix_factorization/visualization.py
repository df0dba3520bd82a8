"""Learning-curve plot with the signature of the reference's
``src/matrix_factorization/visualization.py:6-62`` (matplotlib only; seaborn is not required)."""
import numpy as np


def visualize_learning_curve(training_losses, validation_losses, validation_scores, path=None):
    import matplotlib
    matplotlib.use('Agg')
    import matplotlib.pyplot as plt
    fig, axes = plt.subplots(figsize=(18, 16), nrows=2, dpi=100)
    x = np.arange(1, len(training_losses) + 1)
    axes[0].plot(x, training_losses, '-o', linewidth=2, label='train_loss')
    axes[0].plot(x, validation_losses, '-o', linewidth=2, label='val_loss')
    for name, scores in (validation_scores or {}).items():
        axes[1].plot(x, scores, '-o', linewidth=2, label=name)
    for ax, title in zip(axes, ('Training and Validation Losses', 'Validation Scores')):
        ax.set_xlabel('Epochs/Steps', size=15, labelpad=12.5)
        ax.set_title(title, size=20, pad=15)
        ax.legend(prop={'size': 18})
    if path is None:
        plt.show()
    else:
        plt.savefig(path)
        plt.close(fig)
